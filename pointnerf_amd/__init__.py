"""pointnerf_amd -- MI355X (gfx950) implementation of the Point-NeRF render/optimise hot path.

The product is the C-ABI library ``libpnerf_hip.so`` (sources in ``csrc/``, interface in ``include/pnerf.h``); the
modules of this package mirror the reference's Python interface for that path:

    point_query.lighting_fast_querier / woord_query_grid_point_index   neural-point query
    neural_points.NeuralPoints                                         point cloud module, 14-tuple forward
    point_aggregators.PointAggregator                                  aggregator MLP (reference state_dict keys)
    diff_ray_marching.ray_march, diff_render_func.find_*               renderer
    neural_points_volumetric_model.NeuralPointsRayMarching             the fused hot module
    dist                                                               ray-shard data parallelism (RCCL)
"""
__version__ = "0.1.0"
