"""``models`` package OVERLAY: put this directory's parent FIRST on PYTHONPATH and the reference's scripts run unchanged --

    PYTHONPATH=<repo>/pointnerf_amd/overlay:<repo> POINTNERF_REFERENCE=<Point-NeRF checkout> python <checkout>/run/train_ft.py ...

``from models import create_model`` (run/train_ft.py:12) resolves here; this package is the reference's own ``models`` package
(its __init__ is executed verbatim, its directory is on ``__path__``) except for the hot-path modules, which the overlay
supplies on top of libpnerf_hip.so:

    models/neural_points/point_query.py      lighting_fast_querier, woord_query_grid_point_index
    models/neural_points/neural_points.py    NeuralPoints
    models/aggregators/point_aggregators.py  PointAggregator
    models/rendering/diff_ray_marching.py    ray_march, near_far_linear_ray_generation (the rest: the reference's)
    models/rendering/diff_render_func.py     radiance_render / alpha_blend / tone maps (the rest: the reference's)
    models/neural_points_volumetric_model.py the reference's module with NeuralPointsRayMarching replaced by the fused
                                             render step (PNERF_OVERLAY_FUSED=0 keeps the reference's forward body, which then
                                             runs module by module on the overlay's NeuralPoints / PointAggregator / ray_march)

The model shell (BaseModel, BaseRenderingModel, MvsPointsVolumetricModel: losses, optimizers, schedulers, checkpoints), the
MVSNet initialisation, options, datasets and run scripts are the reference's own, unmodified files.
"""
import os

from ._overlay import REF_MODELS, extend_path

extend_path(__path__)
with open(os.path.join(REF_MODELS, "__init__.py")) as _f:
    exec(compile(_f.read(), os.path.join(REF_MODELS, "__init__.py"), "exec"), globals())
