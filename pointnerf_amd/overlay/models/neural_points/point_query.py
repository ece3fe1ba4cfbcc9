"""models/neural_points/point_query.py of the overlay: the querier of libpnerf_hip.so under the reference's names."""
from pointnerf_amd.point_query import lighting_fast_querier, woord_query_grid_point_index  # noqa: F401
