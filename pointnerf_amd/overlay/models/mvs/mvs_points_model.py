"""models/mvs/mvs_points_model.py of the overlay: the reference's own module (MVSNet, FeatureNet, gen_points, the depth filters: its code,
unmodified) with the two methods that sample the GIVEN feature maps at the candidate points served by libpnerf_hip.so --
``MvsPointsModel.extract_2d`` (:198-218) and ``.query_embedding`` (:225-259) -> pointnerf_amd/mvs_points_model.py (csrc/embed2d.hip)."""
from .._overlay import load_reference_module
from pointnerf_amd import mvs_points_model as _amd

_ref = load_reference_module("mvs/mvs_points_model.py", "models.mvs._reference_mvs_points_model")
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})

import torch as _torch

_ref_extract_2d, _ref_query_embedding = MvsPointsModel.extract_2d, MvsPointsModel.query_embedding      # noqa: F821


def _wants_grad(*ts):
    """the reference's two methods are differentiable (the `feedforward` training scripts send the point features' gradient to FeatureNet through
    gen_points -> query_embedding, mvs_points_model.py:370); the HIP versions sample detached maps and return leaves.  So: whenever autograd is
    recording and any feature map / point position / confidence map carries a gradient, the REFERENCE'S method runs -- training FeatureNet through the
    overlay stays the reference's computation; the one-time point initialisation of the per-scene scripts (no_grad, :370 under torch.no_grad in
    run/train_ft.py) takes the HIP kernels."""
    if not _torch.is_grad_enabled():
        return False
    flat = []
    for t in ts:
        if isinstance(t, (list, tuple)):
            for u in t:
                flat.extend(u if isinstance(u, (list, tuple)) else [u])
        else:
            flat.append(t)
    return any(isinstance(t, _torch.Tensor) and t.requires_grad for t in flat)


def _extract_2d(self, img_feats, view_ids, layer_ids, intrinsics, c2ws, w2cs, cam_xyz, HD, WD, cam_vid=0, **kw):
    if _wants_grad(img_feats, cam_xyz):
        return _ref_extract_2d(self, img_feats, view_ids, layer_ids, intrinsics, c2ws, w2cs, cam_xyz, HD, WD, cam_vid=cam_vid, **kw)
    return _amd.MvsPointsModel.extract_2d(self, img_feats, view_ids, layer_ids, intrinsics, c2ws, w2cs, cam_xyz, HD, WD, cam_vid=cam_vid, **kw)


def _query_embedding(self, HDWD, cam_xyz, photometric_confidence, img_feats, c2ws, w2cs, intrinsics, cam_vid, pointdir_w=False):
    if _wants_grad(img_feats, cam_xyz, photometric_confidence):
        return _ref_query_embedding(self, HDWD, cam_xyz, photometric_confidence, img_feats, c2ws, w2cs, intrinsics, cam_vid, pointdir_w=pointdir_w)
    return _amd.MvsPointsModel.query_embedding(self, HDWD, cam_xyz, photometric_confidence, img_feats, c2ws, w2cs, intrinsics, cam_vid, pointdir_w=pointdir_w)


MvsPointsModel.extract_2d = _extract_2d                                   # noqa: F821
MvsPointsModel.point_dirs = _amd.MvsPointsModel.point_dirs                # noqa: F821
MvsPointsModel.query_embedding = _query_embedding                         # noqa: F821
