"""Parity AT THE CONFIGURATION bench.py TIMES (BASELINE.json configs[1]: 2 M lego points, K = 8, SR = 128, lego script values) on a ray
subsample: neighbor indices bit-exact, sigma / RGB / ray colour <= 1e-4 against the fp32 oracle, and gradients measured against a
FLOAT64 evaluation of the same renderer on the same query result: two fp32 implementations of this network differ from each other
by 1e-5 .. 1e-4 of a gradient tensor's largest element (summation order, and LeakyReLU kinks: a unit whose pre-activation is within
rounding of zero takes the other branch, derivative 1 <-> 0.01), so the meaningful statements are (i) the HIP path is as close to
the exact gradient as the fp32 oracle is, and (ii) every point-gradient element that is off by more than 1e-5 of the tensor's
maximum belongs to a point touched by a neighbor row / sample with such a near-zero pre-activation."""
import numpy as np
import pytest
import torch

from gpu_util import hip_render, DEV
from pointnerf_amd import config, scenes, ops
from oracle import pyref

pytestmark = pytest.mark.gpu
NRAYS = 768


def _bench_case(nrays=NRAYS, first=0):
    opt = config.bench_lego_opt()                   # is_train = 0: no jitter, results are bit-defined
    xyz = torch.from_numpy(scenes.lego_points())
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(xyz.shape[0], 32, 1).items()}
    d = scenes.random_rays(0, 65536)
    # a contiguous run of the step-0 batch that contains hits
    d["raydir"], d["gt_image"], d["pixel_idx"] = d["raydir"][:, first:first + nrays], d["gt_image"][:, first:first + nrays], d["pixel_idx"][:, first:first + nrays]
    inp = pyref.to_torch_inputs(d)
    mlp = pyref.init_mlp_params(opt, seed=0, bias_scale=0.05)
    return opt, xyz, attrs, inp, mlp


def test_bench_config_forward_and_gradients():
    torch.set_num_threads(8)
    opt, xyz, attrs, inp, mlp = _bench_case()
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    op = dict(xyz=xyz, **{k: v.clone().requires_grad_(True) for k, v in attrs.items()})
    ref = pyref.render(opt, op, om, inp, nthreads=8)
    dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp, train=True)
    hit = (dense["ray_hit"] > 0).cpu()
    assert int(hit.sum()) > 50 and torch.equal(hit.to(torch.int8)[None], ref["ray_mask"])
    assert torch.equal(dense["sample_pidx"].cpu()[hit][None], ref["query"]["sample_pidx"])
    errs = {}
    for ours, theirs in [("decoded", "decoded_features"), ("weight", "weight"), ("ray_color", "coarse_raycolor"), ("opacity", "coarse_point_opacity")]:
        a = fwd[ours].cpu()[hit]
        errs[ours] = float((a - ref[theirs][0].detach().reshape(a.shape)).abs().max())
    print("configs[1] forward max abs errors:", errs, "rays hit", int(hit.sum()), "valid samples", ctx["n_valid"])
    assert max(errs.values()) <= 1e-4, errs

    # gradients of a fixed random functional of the ray colours: fp32 oracle, float64 yardstick, HIP path
    probe = torch.rand(ref["coarse_raycolor"].shape, generator=torch.Generator().manual_seed(123))
    (ref["coarse_raycolor"] * probe).sum().backward()
    kink = {}
    out64, p64, m64 = pyref.render_f64(opt, op, om, inp, ref["query"], kink=kink)
    (out64["coarse_raycolor"] * probe.double()).sum().backward()
    # points touched by a row / sample with a pre-activation within EPS of a LeakyReLU kink (float64 pre-activations)
    EPS = 2e-6
    pidx = ref["query"]["sample_pidx"][0]                            # [R'', SR, K]
    mask = pidx >= 0
    row_pts = pidx[mask]                                             # neighbor rows in mask order
    kinked = torch.zeros(xyz.shape[0], dtype=torch.bool)
    kinked[row_pts[kink["row_min_pre"] < EPS].long()] = True
    valid = mask.any(dim=-1)
    smp_pts = pidx[valid][kink["sample_min_pre"] < EPS]              # [n, K]
    kinked[smp_pts[smp_pts >= 0].long()] = True
    touched = int((row_pts.unique() >= 0).sum())
    dev = torch.device(DEV)
    lay, _ = ops.mlp_layout()
    # The backward twice on the same (f16 cross terms) forward: with f16 cross terms in the input-gradient chain (the round-2..5 arithmetic:
    # every bar as it was) and with the shipped e4m3 cross terms (csrc/mixq.h).  The e4m3 chain adds ~1.3e-6 of sum |terms| per 256-term dot
    # product and layer to every d X element, i.e. smooth noise of ~1e-5 of a point tensor's maximum on many points instead of isolated kink
    # flips: the MLP tensors keep their bars; a point gradient may be up to POINT_BAR of the tensor's maximum off WITHOUT a kink explaining it.
    for bits, POINT_BAR in ((16, 1e-5), (8, 1e-4)):
        old_bits, _ = ops.set_cross_terms(bits)
        try:
            if bits != 16:
                dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp, train=True)
            g = torch.zeros(ctx["R"], 3, device=dev)
            g[hit.to(dev)] = probe[0].to(dev)
            gflat = torch.zeros_like(ctx["flat"])
            grads = {k: torch.zeros_like(v) for k, v in ctx["pts_t"].items()}
            ops.render_backward(ctx["cam"], ctx["pts"], ctx["packed"], ctx["flat"], ctx["raydir"], dense, ctx["R"], opt.SR, opt.K,
                                ctx["n_valid"], fwd, g, gflat, grads)
            torch.cuda.synchronize()
        finally:
            ops.set_cross_terms(old_bits)
        print("---- cross terms of the input-gradient chain: %s" % ("f16 (csrc/f16x3.h)" if bits == 16 else "e4m3 (csrc/mixq.h, shipped)"))
        failures = []
        for k, (o, shp) in lay.items():
            ours, o32, g64 = gflat[o:o + int(np.prod(shp))].view(shp).cpu().double(), om[k].grad.double(), m64[k].grad
            scale = float(g64.abs().max())
            e_ours, e_o32 = float((ours - g64).abs().max()) / scale, float((o32 - g64).abs().max()) / scale
            d = (ours - g64).abs()
            rms_ours, rms_o32 = float(d.pow(2).mean().sqrt()) / scale, float((o32 - g64).pow(2).mean().sqrt()) / scale
            frac = float((d > max(3.0 * e_o32, 1e-5) * scale).double().mean())
            print("%-24s max |hip - f64| %.2e  |oracle32 - f64| %.2e   rms %.2e / %.2e   elements beyond max(3 x oracle's, 1e-5): %.1e" %
                  (k, e_ours, e_o32, rms_ours, rms_o32, frac))
            # as close to the exact gradient as the fp32 oracle is, in the mean; isolated kink flips (an output unit's row of dW and its bias
            # entry) bounded by 1e-4 of the tensor's maximum and rare
            # (ONE flipped unit of one row / sample changes that unit's whole row of dW: up to shp[-1] elements, e.g. 280 of the 35 840 of
            #  color_branch.0.weight = 0.8 % -- the count allowance is therefore at least one row)
            cnt_ok = frac * d.numel() <= max(2.0, 2e-3 * d.numel(), float(shp[-1]))
            if not (rms_ours <= max(4.0 * rms_o32, 5e-6) and e_ours <= max(3.0 * e_o32, 1e-4) and cnt_ok):
                failures.append((k, e_ours, e_o32, rms_ours, rms_o32, frac))
        assert not failures, failures
        n_bad_total = 0
        for k in ("points_embeding", "points_conf", "points_color", "points_dir"):
            a, b, o32 = grads[k].cpu().double(), p64[k].grad[0], op[k].grad[0].double()
            e = (a - b).abs()
            scale = float(b.abs().max())
            bad = (e > POINT_BAR * scale).any(dim=-1)
            bad5 = (e > 1e-5 * scale).any(dim=-1)
            bad32 = ((o32 - b).abs() > 1e-5 * scale).any(dim=-1)
            n_bad_total += int(bad.sum())
            nk = ~kinked
            print("%-18s max|grad| %.3e  |hip - f64| %.2e (points no kink touches: %.2e)  |oracle32 - f64| %.2e  points beyond 1e-5 max: hip %d, oracle32 %d; beyond %.0e: %d (all kink-attributed: %s)" %
                  (k, scale, float(e.max()) / scale, float(e[nk].max()) / scale, float((o32 - b).abs().max()) / scale, int(bad5.sum()), int(bad32.sum()), POINT_BAR, int(bad.sum()), bool(kinked[bad].all())))
            assert bool(kinked[bad].all()), (k, "out-of-tolerance gradient on a point no kink explains", bad.nonzero()[:5].tolist())
            assert float(e.max()) <= 2e-2 * scale, k
        print("points touched %d, kink-affected %d, with an out-of-tolerance element %d" % (touched, int(kinked.sum()), n_bad_total))
        assert n_bad_total <= max(4, touched // 500)


WG_RAYS, WG_CHUNK = 8192, 1024


def test_bench_config_weight_gradients_8192_rays():
    """The weight gradients of the timed configuration on 8 192 rays of the step-0 batch (round 3 measured 768), against FLOAT64, for the
    three arithmetics that exist: the fp32 oracle (what the reference's cuBLAS SGEMMs compute, up to summation order), the shipped
    one-f16-plane operands (ops.set_wgrad_planes(1)) and the two-plane / three-product mode (set_wgrad_planes(2): fp32-class).  The float64
    and fp32 oracle gradients are accumulated over chunks of 1 024 rays (a gradient of a sum); the device runs the 8 192 rays as one step.
    Bars (per MLP tensor, relative to the tensor's largest |dW| in float64):
      * the eleven tensors the weight-gradient GEMMs produce (dW of the four aggregator and the first three colour layers, the aggregator
        biases = the GEMMs' ones column), two planes: rms error <= max(2 x the fp32 oracle's own, 5e-7) -- the same class of arithmetic
        (measured 1.4e-7 .. 4.3e-7 against the oracle's 0.8e-7 .. 3.5e-7);
      * every tensor, one plane: rms error <= 6e-6 -- the random-walk budget of DESIGN 4.1 (1.6e-4 sqrt(sum t^2) per element; 4.7e-6 rms for a
        structured gradient in tests/test_split_f16_cpu.py), which does NOT depend on what the oracle's own error happens to be (measured: 1.9e-6 on
        layer 1, <= 7e-7 on the other GEMM tensors; the heads and colour biases are sums formed by the tile kernels, the same in both modes);
      * max error of both <= max(3 x the oracle's, 1e-4): measured 3.6e-5 on layer 1 with one plane (two planes: 3.1e-6 = the oracle's)."""
    torch.set_num_threads(8)
    opt, xyz, attrs, inp, mlp = _bench_case(WG_RAYS)
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    op = dict(xyz=xyz, **{k: v.clone().requires_grad_(True) for k, v in attrs.items()})
    m64 = {k: torch.zeros(v.shape, dtype=torch.float64) for k, v in mlp.items()}
    probes, hits = [], 0
    gen = torch.Generator().manual_seed(123)
    for c0 in range(0, WG_RAYS, WG_CHUNK):
        ci = dict(inp)
        for k in ("raydir", "gt_image", "pixel_idx"):
            ci[k] = inp[k][:, c0:c0 + WG_CHUNK]
        ref = pyref.render(opt, op, om, ci, nthreads=8)
        probe = torch.rand(ref["coarse_raycolor"].shape, generator=gen)
        probes.append(probe[0])
        hits += probe.shape[1]
        if probe.shape[1] == 0:
            continue
        (ref["coarse_raycolor"] * probe).sum().backward()                      # accumulates into om[k].grad
        out64, _, mm = pyref.render_f64(opt, op, om, ci, ref["query"])
        (out64["coarse_raycolor"] * probe.double()).sum().backward()
        for k in m64:
            m64[k] += mm[k].grad
    probe_all = torch.cat(probes, 0)
    dev = torch.device(DEV)
    lay, _ = ops.mlp_layout()
    res = {}
    for planes in (1, 2):
        old = ops.set_wgrad_planes(planes)
        try:
            dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp, train=True)
            hit = dense["ray_hit"] > 0
            assert int(hit.sum()) == hits
            g = torch.zeros(ctx["R"], 3, device=dev)
            g[hit] = probe_all.to(dev)
            gflat = torch.zeros_like(ctx["flat"])
            grads = {k: torch.zeros_like(v) for k, v in ctx["pts_t"].items()}
            ops.render_backward(ctx["cam"], ctx["pts"], ctx["packed"], ctx["flat"], ctx["raydir"], dense, ctx["R"], opt.SR, opt.K,
                                ctx["n_valid"], fwd, g, gflat, grads)
            torch.cuda.synchronize()
            res[planes] = {k: gflat[o:o + int(np.prod(shp))].view(shp).cpu().double() for k, (o, shp) in lay.items()}
        finally:
            ops.set_wgrad_planes(old)
    print("rays %d, hit %d, valid samples %d; errors relative to max |dW| (float64)" % (WG_RAYS, hits, ctx["n_valid"]))
    print("%-24s %-23s %-23s %-23s" % ("tensor", "fp32 oracle rms / max", "one plane rms / max", "two planes rms / max"))
    failures = []
    for k in lay:
        g64 = m64[k]
        scale = float(g64.abs().max())
        st = {}
        for name, t in (("o32", om[k].grad.double()), ("p1", res[1][k]), ("p2", res[2][k])):
            d = (t - g64).abs()
            st[name] = (float(d.pow(2).mean().sqrt()) / scale, float(d.max()) / scale)
        print("%-24s %.2e / %.2e     %.2e / %.2e     %.2e / %.2e" % (k, *st["o32"], *st["p1"], *st["p2"]))
        gemm = k.startswith("block") or k in ("color_branch.0.weight", "color_branch.2.weight", "color_branch.4.weight")
        ok = st["p1"][0] <= 6e-6 and st["p2"][0] <= 6e-6 and max(st["p1"][1], st["p2"][1]) <= max(3.0 * st["o32"][1], 1e-4)
        if gemm:
            ok = ok and st["p2"][0] <= max(2.0 * st["o32"][0], 5e-7)
        if not ok:
            failures.append((k, st))
    assert not failures, failures
