"""The HIP kernels executed on the HOST (tools/emu: same sources, every GPU thread a fiber, MFMA / shuffles / barriers
emulated) against the CPU oracle.  These run in the CPU suite: a kernel indexing, layout or synchronisation bug fails here,
before any GPU time is spent; the -m gpu tests remain the parity tests proper (the emulator's MFMA accumulation order and
transcendental functions are the host's, so the tolerances here are the same bars, not tighter ones)."""
import numpy as np
import pytest
import torch

import test_gpu_backward as TB
import test_gpu_render as TR
from cases import build_case
from emu_util import emu_backend
from pointnerf_amd import config, scenes
from oracle import pyref


@pytest.fixture(autouse=True)
def _emu(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(TB, "DEV", "cpu")
    with emu_backend():
        yield


def _tiny_case(K, SR, size, n=1200, seed=5):
    opt = config.lego_opt(K=K, SR=SR, P=24, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    xyz = torch.from_numpy(scenes.chair_points(n, seed=seed, radius=0.06))
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, 32, seed).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=55.0, x0=400 - size // 2, y0=400 - size // 2, size=size))
    mlp = pyref.init_mlp_params(opt, seed=3, bias_scale=0.1)
    return opt, xyz, attrs, inp, mlp


def test_emulated_forward_matches_oracle():
    TR._compare(*build_case("small_k4"))


@pytest.mark.parametrize("K,SR,size", [(8, 12, 5), (3, 10, 4)])
def test_emulated_forward_and_backward_match_oracle(K, SR, size):
    case = _tiny_case(K, SR, size)
    TR._compare(*case)
    TB._run(*case)
