"""models/rendering/diff_ray_marching.py of the overlay: every name of the reference's module, with ray_march and
near_far_linear_ray_generation replaced by the HIP-backed ones."""
from .._overlay import load_reference_module

_ref = load_reference_module("rendering/diff_ray_marching.py", "models.rendering._reference_diff_ray_marching")
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})

from pointnerf_amd.diff_ray_marching import ray_march, near_far_linear_ray_generation  # noqa: E402,F401
