"""GPU parity of the HIP aggregator MLP + colour MLP + ray-march forward (through the C ABI) against the
CPU oracle (oracle/pyref.py) and against the golden vectors of the reference's own modules.
Tolerance: 1e-4 absolute on sigma / RGB / ray colour (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from cases import CASES, build_case
from gpu_util import hip_render
from pointnerf_amd import config, scenes
from oracle import pyref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def _compare(opt, xyz, attrs, inp, mlp, fix=None):
    torch.set_num_threads(8)
    points = dict(xyz=xyz, **attrs)
    with torch.no_grad():
        ref = pyref.render(opt, points, mlp, inp, nthreads=8)
    dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp)
    hit = (dense["ray_hit"] > 0).cpu()
    assert torch.equal(hit.to(torch.int8)[None], ref["ray_mask"])
    errs = {}
    for ours, theirs in [("decoded", "decoded_features"), ("weight", "weight"), ("ray_color", "coarse_raycolor"),
                         ("opacity", "coarse_point_opacity"), ("bg_trans", "coarse_is_background"), ("blend_w", "blend_weight")]:
        a = fwd[ours].cpu()[hit]
        b = ref[theirs][0].reshape(a.shape)
        errs[ours] = float((a - b).abs().max())
    print("max abs errors:", errs)
    for k, v in errs.items():
        assert v <= TOL, (k, v)
    # rays that hit nothing: background colour, full transmittance
    miss = ~hit
    if miss.any():
        assert float((fwd["ray_color"].cpu()[miss] - inp["bg_color"][0]).abs().max()) <= 1e-6
        assert float((fwd["bg_trans"].cpu()[miss] - 1).abs().max()) <= 1e-6
    if fix is not None:
        assert np.abs(fwd["decoded"].cpu()[hit].numpy() - fix["output"][0]).max() <= TOL
        assert np.abs(fwd["ray_color"].cpu()[hit].numpy() - fix["ray_color"][0]).max() <= TOL
    return errs


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_oracle_and_reference_golden(name):
    opt, xyz, attrs, inp, mlp = build_case(name)
    _compare(opt, xyz, attrs, inp, mlp, fix=np.load(os.path.join(G, "agg_%s.npz" % name)))


def test_forward_config1_chair():
    opt = config.chair_opt()
    xyz = torch.from_numpy(scenes.chair_points())
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(8192, 32, 0).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays())
    mlp = pyref.init_mlp_params(opt, seed=0, bias_scale=0.05)
    _compare(opt, xyz, attrs, inp, mlp)


@pytest.mark.parametrize("K,SR", [(12, 20), (1, 8), (16, 12), (6, 70)])
def test_forward_other_K(K, SR):
    opt = config.lego_opt(K=K, SR=SR, P=24, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    xyz = torch.from_numpy(scenes.chair_points(2500, seed=5, radius=0.06))
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(2500, 32, 5).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=55.0, x0=394, y0=394, size=12))
    mlp = pyref.init_mlp_params(opt, seed=3, bias_scale=0.1)
    _compare(opt, xyz, attrs, inp, mlp)
