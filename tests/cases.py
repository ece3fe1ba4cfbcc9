"""Seeded parity cases shared by tests/golden/make_golden.py and the tests (inputs are regenerated,
never stored)."""
import torch

from pointnerf_amd import config, scenes
from oracle import pyref

CASES = {
    # name: (opt overrides, n_points, ray block size, seed)
    "small_k8": (dict(K=8, SR=24, P=12, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3]), 1500, 12, 0),
    "small_k4": (dict(K=4, SR=16, P=12, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3]), 900, 10, 1),
}


def build_case(name):
    ov, n, size, seed = CASES[name]
    opt = config.lego_opt(**ov)
    xyz = torch.from_numpy(scenes.chair_points(n, seed=seed, radius=0.06))
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, opt.point_features_dim, seed).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=30.0 + 40 * seed, x0=400 - size // 2,
                                                  y0=400 - size // 2, size=size))
    mlp = pyref.init_mlp_params(opt, seed=seed, bias_scale=0.1)
    return opt, xyz, attrs, inp, mlp


def probe_scalar(color, conf_c):
    """The fixed scalar whose gradients the golden files pin."""
    g = torch.Generator().manual_seed(123)
    probe = torch.rand(color.shape, generator=g)
    v = conf_c.clamp(1e-3, 1 - 1e-3)
    return (color * probe.to(color.device)).sum() + 0.05 * torch.mean(torch.log(v) + torch.log(1 - v))
