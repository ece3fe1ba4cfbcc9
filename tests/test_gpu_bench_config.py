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


def _bench_case():
    opt = config.bench_lego_opt()                   # is_train = 0: no jitter, results are bit-defined
    xyz = torch.from_numpy(scenes.lego_points())
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(xyz.shape[0], 32, 1).items()}
    d = scenes.random_rays(0, 65536)
    # a contiguous run of the step-0 batch that contains hits
    d["raydir"], d["gt_image"], d["pixel_idx"] = d["raydir"][:, :NRAYS], d["gt_image"][:, :NRAYS], d["pixel_idx"][:, :NRAYS]
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
    if max(errs.values()) > 1e-4:          # say WHERE: sample, channel, its neighbor count (sample class), both values
        a = fwd["decoded"].cpu()[hit].reshape(-1, 4)
        b = ref["decoded_features"][0].detach().reshape(-1, 4)
        nn = (dense["sample_pidx"].cpu()[hit].reshape(-1, opt.K) >= 0).sum(-1)
        bad = torch.nonzero((a - b).abs().amax(-1) > 1e-4)[:, 0]
        errs["where"] = [(int(i), int(i) // opt.SR, int(i) % opt.SR, int(nn[i]), a[i].tolist(), b[i].tolist()) for i in bad[:6]]
        errs["n_bad"] = int(bad.numel())
        # which side moved?  run both again on the same inputs
        ref2 = pyref.render(opt, op, om, inp, nthreads=8)
        _, fwd2, _ = hip_render(opt, xyz, attrs, inp, mlp, train=True)
        a2 = fwd2["decoded"].cpu()[hit].reshape(-1, 4)
        b2 = ref2["decoded_features"][0].detach().reshape(-1, 4)
        errs["second_run"] = [(int(i), a2[i].tolist(), b2[i].tolist()) for i in bad[:6]]
        errs["hip_runs_equal"] = bool(torch.equal(a, a2))
        errs["oracle_runs_equal"] = bool(torch.equal(b, b2))
        import json, os
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/forward_mismatch.jsonl", "a") as fh:
            fh.write(json.dumps(errs) + "\n")
    assert max(v for k, v in errs.items() if k not in ("where", "n_bad", "second_run", "hip_runs_equal", "oracle_runs_equal")) <= 1e-4, errs

    # gradients of a fixed random functional of the ray colours: fp32 oracle, float64 yardstick, HIP path
    probe = torch.rand(ref["coarse_raycolor"].shape, generator=torch.Generator().manual_seed(123))
    (ref["coarse_raycolor"] * probe).sum().backward()
    kink = {}
    out64, p64, m64 = pyref.render_f64(opt, op, om, inp, ref["query"], kink=kink)
    (out64["coarse_raycolor"] * probe.double()).sum().backward()
    dev = torch.device(DEV)
    g = torch.zeros(ctx["R"], 3, device=dev)
    g[hit.to(dev)] = probe[0].to(dev)
    gflat = torch.zeros_like(ctx["flat"])
    grads = {k: torch.zeros_like(v) for k, v in ctx["pts_t"].items()}
    ops.render_backward(ctx["cam"], ctx["pts"], ctx["packed"], ctx["flat"], ctx["raydir"], dense, ctx["R"], opt.SR, opt.K,
                        ctx["n_valid"], fwd, g, gflat, grads)
    torch.cuda.synchronize()
    lay, _ = ops.mlp_layout()
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
        if not (rms_ours <= max(4.0 * rms_o32, 5e-6) and e_ours <= max(3.0 * e_o32, 1e-4) and frac * d.numel() <= max(2.0, 2e-3 * d.numel(), float(shp[-1]))):
            failures.append((k, e_ours, e_o32, rms_ours, rms_o32, frac))
    assert not failures, failures

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
    n_bad_total = 0
    for k in ("points_embeding", "points_conf", "points_color", "points_dir"):
        a, b, o32 = grads[k].cpu().double(), p64[k].grad[0], op[k].grad[0].double()
        e = (a - b).abs()
        scale = float(b.abs().max())
        bad = (e > 1e-5 * scale).any(dim=-1)
        bad32 = ((o32 - b).abs() > 1e-5 * scale).any(dim=-1)
        n_bad_total += int(bad.sum())
        print("%-18s max|grad| %.3e  |hip - f64| %.2e  |oracle32 - f64| %.2e  points beyond 1e-5 max: hip %d, oracle32 %d (hip's all kink-attributed: %s)" %
              (k, scale, float(e.max()) / scale, float((o32 - b).abs().max()) / scale, int(bad.sum()), int(bad32.sum()), bool(kinked[bad].all())))
        assert bool(kinked[bad].all()), (k, "out-of-tolerance gradient on a point no kink explains", bad.nonzero()[:5].tolist())
        assert float(e.max()) <= 2e-2 * scale, k
    touched = int((row_pts.unique() >= 0).sum())
    print("points touched %d, kink-affected %d, with an out-of-tolerance element %d" % (touched, int(kinked.sum()), n_bad_total))
    assert n_bad_total <= max(4, touched // 500)
