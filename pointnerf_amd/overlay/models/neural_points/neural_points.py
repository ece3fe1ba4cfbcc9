"""models/neural_points/neural_points.py of the overlay: NeuralPoints on libpnerf_hip.so.  The command-line options are the
reference's own (its static option table is taken from the reference's class, so names and defaults cannot drift)."""
from .._overlay import load_reference_module
from pointnerf_amd.neural_points import NeuralPoints as _NeuralPoints

_ref = load_reference_module("neural_points/neural_points.py", "models.neural_points._reference_neural_points")


class NeuralPoints(_NeuralPoints):
    modify_commandline_options = staticmethod(_ref.NeuralPoints.modify_commandline_options)
