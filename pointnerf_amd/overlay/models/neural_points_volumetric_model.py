"""models/neural_points_volumetric_model.py of the overlay: the reference's module, executed verbatim; unless
PNERF_OVERLAY_FUSED=0 its NeuralPointsRayMarching (the network the model shell builds, :165-168, and calls, :214) is the fused
render step of pointnerf_amd -- same constructor keywords, same forward keywords, same output dictionary."""
import os

from ._overlay import load_reference_module

_ref = load_reference_module("neural_points_volumetric_model.py", "models._reference_neural_points_volumetric_model")
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})

if os.environ.get("PNERF_OVERLAY_FUSED", "1") != "0":
    from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching  # noqa: E402,F401
    _ref.NeuralPointsRayMarching = NeuralPointsRayMarching          # create_network_models looks the name up in its own module
