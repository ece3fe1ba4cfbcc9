"""Drop-in for ``models/rendering/diff_render_func.py`` (:8-67): the lookup functions the model shell calls
(``base_rendering_model.py:314-337``) and the three functions every script selects (radiance / alpha / off).  The
functions are tiny tensor expressions, kept so that reference code that calls them directly still works; the fused
renderer (``diff_ray_marching.ray_march``) recognises them by identity and runs them inside its HIP kernel."""
import torch
import torch.nn.functional as F


def radiance_render(ray_feature):
    return ray_feature[..., 1:4]


def white_color(ray_feature):
    albedo = ray_feature[..., 1:4].clamp(0., 1.)
    return torch.ones_like(albedo)


def alpha_blend(opacity, acc_transmission):
    return opacity * acc_transmission


def alpha2_blend(opacity, acc_transmission):
    return opacity * acc_transmission * acc_transmission


def simple_tone_map(color, gamma=2.2, exposure=1):
    return torch.pow(color * exposure + 1e-5, 1 / gamma).clamp_(0, 1)


def no_tone_map(color, gamma=2.2, exposure=1):
    return color


def normalize_tone_map(color):
    return F.normalize(color, dim=-1) * 0.5 + 0.5


def find_render_function(name):
    if name == 'radiance':
        return radiance_render
    elif name == 'white':
        return white_color
    raise RuntimeError('Unknown render function: ' + name)


def find_blend_function(name):
    if name == 'alpha':
        return alpha_blend
    elif name == 'alpha2':
        return alpha2_blend
    raise RuntimeError('Unknown blend function: ' + name)


def find_tone_map(name):
    if name == 'gamma':
        return simple_tone_map
    elif name == 'normalize':
        return normalize_tone_map
    elif name == 'off':
        return no_tone_map
    raise RuntimeError('Unknown blend function: ' + name)
