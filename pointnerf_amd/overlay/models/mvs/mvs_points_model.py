"""models/mvs/mvs_points_model.py of the overlay: the reference's own module (MVSNet, FeatureNet, gen_points, the depth filters: its code,
unmodified) with the two methods that sample the GIVEN feature maps at the candidate points served by libpnerf_hip.so --
``MvsPointsModel.extract_2d`` (:198-218) and ``.query_embedding`` (:225-259) -> pointnerf_amd/mvs_points_model.py (csrc/embed2d.hip)."""
from .._overlay import load_reference_module
from pointnerf_amd import mvs_points_model as _amd

_ref = load_reference_module("mvs/mvs_points_model.py", "models.mvs._reference_mvs_points_model")
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})

MvsPointsModel.extract_2d = _amd.MvsPointsModel.extract_2d                # noqa: F821
MvsPointsModel.point_dirs = _amd.MvsPointsModel.point_dirs                # noqa: F821
MvsPointsModel.query_embedding = _amd.MvsPointsModel.query_embedding      # noqa: F821
