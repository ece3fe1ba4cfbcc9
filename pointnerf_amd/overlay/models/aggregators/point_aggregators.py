"""models/aggregators/point_aggregators.py of the overlay: PointAggregator on libpnerf_hip.so, with the reference's own
command-line option table."""
from .._overlay import load_reference_module
from pointnerf_amd.point_aggregators import PointAggregator as _PointAggregator

_ref = load_reference_module("aggregators/point_aggregators.py", "models.aggregators._reference_point_aggregators")


class PointAggregator(_PointAggregator):
    modify_commandline_options = staticmethod(_ref.PointAggregator.modify_commandline_options)
