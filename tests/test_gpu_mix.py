"""The mixed-format tile GEMM of csrc/mixq.h on the device (f16 h.h + e4m3 cross terms on v_mfma_scale_f32_32x32x64_f8f6f4): one 64-row tile
against the numpy restatement of the format (layouts, block scales, saturation: to fp32 accumulation noise) and against float64 (the error the
format costs: rms <= 2e-6, max <= 1e-5 of sum |terms|, where three f16 products give 3e-8 and the bar on sigma / RGB downstream is 1e-4), and the
switch back to f16 cross terms (pnerf_set_cross_terms)."""
import ctypes

import numpy as np
import pytest
import torch

import mix_case
from gpu_util import DEV
from pointnerf_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [256, 272, 288])
def test_mixed_tile_gemm_against_restatement_and_float64(K):
    dev = torch.device(DEV)
    lib = L.lib()
    for big in (False, True):
        x, w = mix_case.build(K, big=big)
        dx, dw = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
        img = torch.zeros(lib.pnerf_mlp_packed_bytes(), dtype=torch.uint8, device=dev)
        out = torch.full((64, 256), float("nan"), device=dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        L.check(lib.pnerf_debug_mix_gemm(P(dw), K, P(dx), P(img), P(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnerf_debug_mix_gemm")
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype(np.float64)
        ref = mix_case.restate(x, w)
        ex, sab = mix_case.exact(x, w)
        # (the e4m3 MFMA adds its 64 products with less than fp32 precision inside: measured 1.5e-5 of one instruction's sum, i.e. 1e-8 of the result)
        assert np.abs(o - ref).max() <= 5e-6 * np.abs(ref).max(), np.abs(o - ref).max() / np.abs(ref).max()
        e = (o - ex) / sab
        print("K", K, "big", big, "device vs float64: rms %.2e max %.2e of sum|terms|;  vs restatement %.1e" % (np.sqrt((e ** 2).mean()), np.abs(e).max(), np.abs(o - ref).max() / np.abs(ref).max()))
        assert np.abs(e).max() <= 1e-5 and np.sqrt((e ** 2).mean()) <= 2e-6


def test_mixed_tile_gemm_saturates_instead_of_poisoning():
    """|x| beyond the e4m3 range of a slot (448 for h, ~900 for the residual) and beyond f16 (65504): the cross term degrades, nothing becomes NaN"""
    dev = torch.device(DEV)
    lib = L.lib()
    x, w = mix_case.build(256)
    x[3, 17] = 700.0; x[5, 100] = -3000.0; x[7, 200] = 1.0e5
    dx, dw = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    img = torch.zeros(lib.pnerf_mlp_packed_bytes(), dtype=torch.uint8, device=dev)
    out = torch.full((64, 256), float("nan"), device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(lib.pnerf_debug_mix_gemm(P(dw), 256, P(dx), P(img), P(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnerf_debug_mix_gemm")
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.float64)
    assert np.isfinite(o).all()
    xc = np.clip(x, -65504, 65504)
    ex, sab = mix_case.exact(xc, w)
    e = np.abs(o - ex) / sab
    print("rows with out-of-range values: max err / sum|terms| %.2e (rows 3, 5), %.2e (row 7: clamped to 65504)" % (e[[3, 5]].max(), e[7].max()))
    assert e[[3, 5]].max() <= 3e-4 and e[7].max() <= 3e-4          # the outlier's own cross term is lost (2^-12 of its product), nothing else
    keep = np.ones(64, bool); keep[[3, 5, 7]] = False
    assert e[keep].max() <= 1e-5
