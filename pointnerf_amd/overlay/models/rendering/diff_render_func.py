"""models/rendering/diff_render_func.py of the overlay: the reference's module, with the functions the HIP ray-marcher
recognises (radiance_render, alpha_blend, the look-ups that return them) taken from pointnerf_amd."""
from .._overlay import load_reference_module

_ref = load_reference_module("rendering/diff_render_func.py", "models.rendering._reference_diff_render_func")
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})

from pointnerf_amd.diff_render_func import *  # noqa: E402,F401,F403
