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
    assert e[[3, 5]].max() <= 6e-4 and e[7].max() <= 6e-4          # the outlier's own cross terms are lost (2 x 2^-12 of its product, which dominates sum |terms|), nothing else
    keep = np.ones(64, bool); keep[[3, 5, 7]] = False
    assert e[keep].max() <= 1e-5


# ---- the OPTIONAL forward modes (pnerf_set_cross_terms_where bits 0 / 1: e4m3 cross terms in the inference / training forward; the default keeps f16
# cross terms there): sigma / RGB / ray colour inside north_star's 1e-4 at configs[0], configs[1] and at trained-magnitude embeddings, figures printed
def _forward_errors(opt, xyz, attrs, inp, mlp, train):
    from gpu_util import hip_render
    from oracle import pyref
    from pointnerf_amd import ops
    with torch.no_grad():
        ref = pyref.render(opt, dict(xyz=xyz, **attrs), mlp, inp, nthreads=8)
    out = {}
    for tag, where in (("f16", 4), ("e4m3", 7)):
        old = ops.set_cross_terms(8, where=where)
        try:
            dense, fwd, _ = hip_render(opt, xyz, attrs, inp, mlp, train=train)
            torch.cuda.synchronize()
        finally:
            ops.set_cross_terms(8, where=old[1])
        hit = (dense["ray_hit"] > 0).cpu()
        assert torch.equal(dense["sample_pidx"].cpu()[hit][None], ref["query"]["sample_pidx"])
        dec, rv, d_ref = fwd["decoded"].cpu()[hit], ref["ray_valid"][0], ref["decoded_features"][0]
        out[tag] = (float((dec[..., 0] - d_ref[..., 0])[rv].abs().max()), float((dec[..., 1:] - d_ref[..., 1:])[rv].abs().max()),
                    float((fwd["ray_color"].cpu()[hit] - ref["coarse_raycolor"][0]).abs().max()), float(d_ref[..., 0].abs().max()))
    return out


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("name", ["small_k8", "small_k4", "chair", "lego", "trained_k8"])
def test_forward_with_e4m3_cross_terms_stays_inside_the_bar(name, train):
    from cases import build_case
    from pointnerf_amd import config, scenes
    from oracle import pyref
    torch.set_num_threads(8)
    if name == "chair":                 # BASELINE.json configs[0]
        opt = config.chair_opt()
        xyz = torch.from_numpy(scenes.chair_points())
        attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(8192, 32, 0).items()}
        inp = pyref.to_torch_inputs(scenes.block_rays())
        mlp = pyref.init_mlp_params(opt, seed=0, bias_scale=0.05)
    elif name == "lego":                # configs[1], the timed configuration, 768-ray subsample
        import test_gpu_bench_config as TBC
        opt, xyz, attrs, inp, mlp = TBC._bench_case()
    elif name == "trained_k8":          # embeddings ~N(0, 3^2), |e| <= 10 (tests/test_gpu_trig.py)
        import test_gpu_trig as TT
        opt, xyz, attrs, inp, mlp = TT._big_embedding_case("small_k8")
    else:
        opt, xyz, attrs, inp, mlp = build_case(name)
    e = _forward_errors(opt, xyz, attrs, inp, mlp, train)
    print("%s train=%s (sigma err, rgb err, ray colour err, max sigma): f16 cross terms %s   e4m3 cross terms %s" % (name, train, e["f16"], e["e4m3"]))
    s, r, c, smax = e["e4m3"]
    assert r <= 1e-4 and c <= 1e-4 and s <= 1e-4 * max(1.0, smax)
