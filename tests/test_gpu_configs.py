"""BASELINE.json configs[3] (ScanNet-scale, 6M points, K=8, SR=160) and configs[4] (Barn-scale, 20M points, K=12,
SR=128) as parity-test cases on the GPU, on a ray subsample against the oracle: neighbor indices / sample locations / ray mask
bit-exact; per-sample sigma and RGB (`decoded`), the aggregator weights, the opacities and the ray colours within 1e-4
(north_star's bar; models/aggregators/point_aggregators.py:608-628, models/rendering/diff_ray_marching.py:508-554); the gradients
of a fixed probe functional of the ray colours inside the bars of tests/test_gpu_backward.py (K = 12 runs the 12 / 6 / 3 sample
classes, P = 30 cells at ScanNet scale).  Then a full-size forward + backward with sanity properties."""
import numpy as np
import pytest
import torch

import test_gpu_backward as TB
from gpu_util import DEV, hip_render
from pointnerf_amd import config, scenes, ops
from pointnerf_amd.neural_points import NeuralPoints
from pointnerf_amd.point_aggregators import PointAggregator
from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
from oracle import pyref

pytestmark = pytest.mark.gpu


def _run(opt, xyz_np, ray_fn, n_sub, n_full, seed):
    dev = torch.device(DEV)
    xyz = torch.from_numpy(xyz_np)
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(xyz.shape[0], 32, seed).items()}
    mlp = pyref.init_mlp_params(opt, seed=seed, bias_scale=0.05)
    agg = PointAggregator(opt).to(dev)
    agg.load_state_dict(mlp)
    agg.flatten_()
    npnt = NeuralPoints(32, xyz.shape[0], opt, dev)
    a = {k: v.to(dev) for k, v in attrs.items()}
    npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"],
                    points_conf=a["points_conf"], parameter=True)
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    # --- subsample parity against the oracle: the C-ABI ops directly (query, training forward, backward), then the model's forward
    torch.set_num_threads(8)
    inp = pyref.to_torch_inputs(ray_fn(2, n_sub))
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    op = dict(xyz=xyz, **{k: v.clone().requires_grad_(True) for k, v in attrs.items()})
    ref = pyref.render(opt, op, om, inp, nthreads=8)
    assert ref["coarse_raycolor"].shape[1] > 0
    dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp, train=True)
    hit = (dense["ray_hit"] > 0).cpu()
    assert torch.equal(hit.to(torch.int8)[None], ref["ray_mask"])
    assert torch.equal(dense["sample_pidx"].cpu()[hit][None], ref["query"]["sample_pidx"])
    assert torch.equal(dense["sample_loc"].cpu()[hit][None], ref["query"]["sample_loc_w"])
    errs = {}
    for ours, theirs in [("decoded", "decoded_features"), ("weight", "weight"), ("ray_color", "coarse_raycolor"), ("opacity", "coarse_point_opacity")]:
        a_ = fwd[ours].cpu()[hit]
        errs[ours] = float((a_ - ref[theirs][0].detach().reshape(a_.shape)).abs().max())
    print("forward max abs errors (sigma | RGB per sample, weights, ray colour, opacity):", errs, "rays hit", int(hit.sum()), "valid samples", ctx["n_valid"])
    assert max(errs.values()) <= 1e-4, errs
    probe = torch.rand(ref["coarse_raycolor"].shape, generator=torch.Generator().manual_seed(123))
    (ref["coarse_raycolor"] * probe).sum().backward()
    g = torch.zeros(ctx["R"], 3, device=dev)
    g[hit.to(dev)] = probe[0].to(dev)
    gflat = torch.zeros_like(ctx["flat"])
    grads = {k: torch.zeros_like(v) for k, v in ctx["pts_t"].items()}
    ops.render_backward(ctx["cam"], ctx["pts"], ctx["packed"], ctx["flat"], ctx["raydir"], dense, ctx["R"], opt.SR, opt.K,
                        ctx["n_valid"], fwd, g, gflat, grads)
    torch.cuda.synchronize()
    lay, _ = ops.mlp_layout()
    for k, (o, shp) in lay.items():
        TB._check(k, gflat[o:o + int(np.prod(shp))].view(shp).cpu(), om[k].grad)
    # (point tensors: the fraction criterion of tests/test_gpu_backward.py as it is -- 99.8 % of the elements within 2e-4 of the tensor's maximum; the
    #  cap on a single element is tests/test_gpu_bench_config.py's 2e-2: among 6 M / 20 M points x 32 columns ONE LeakyReLU kink flip against the
    #  oracle shows as 6.5e-3 of the maximum, with f16 and with e4m3 cross terms alike: profiles/r06_configs_gradients.txt)
    for k in attrs:
        TB._check(k, grads[k].cpu(), op[k].grad[0], point_cap=2e-2)
    del op, om, grads, gflat, fwd, dense, ctx
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    with torch.no_grad():
        out = model(**d)
    assert torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])
    hit = npnt.querier.last_dense["ray_hit"].cpu() > 0
    assert torch.equal(npnt.querier.last_dense["sample_pidx"].cpu()[hit][None], ref["query"]["sample_pidx"])
    err = float((out["coarse_raycolor"].cpu() - ref["coarse_raycolor"].detach()).abs().max())
    assert err <= 1e-4, err
    # --- full-size forward + backward
    inp = pyref.to_torch_inputs(ray_fn(5, n_full))
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    # the forward has no atomics: the same bits launch after launch (tests/test_gpu_reproducible.py does this at configs[1]; here the
    # K = 8 one-pass tail at ScanNet scale and the K = 12 / 6 / 3 three-pass tail at Barn scale), inference and training forward
    first = None
    for it in range(60):
        with torch.set_grad_enabled(it % 3 == 2):
            o = model(**d)
        cur = (o["coarse_raycolor"].detach().clone(), o["coarse_point_opacity"].detach().clone())
        if first is None:
            first = cur
        assert all(torch.equal(a, b) for a, b in zip(cur, first)), "forward %d differs from the first one" % it
    out = model(**d)
    loss = pyref.training_loss(opt, out, {"gt_image": d["gt_image"]})
    loss.backward()
    st = model.last_stats
    assert st["rays_hit"] > 0 and st["n_neighbor_rows"] <= st["n_valid_samples"] * opt.K
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in agg.parameters())
    g = npnt.points_embeding.grad
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    col = out["coarse_raycolor"]
    assert float(col.min()) >= -1e-3 and float(col.max()) <= 1.0 + 2e-3
    return st, err


def test_config3_scannet_scale():
    st, err = _run(config.scannet_opt(), scenes.scannet_points(), scenes.scannet_rays, 384, 16384, 3)
    print("scannet", st, "colour err", err)


def test_config4_barn_scale_k12():
    st, err = _run(config.barn_opt(), scenes.barn_points(), scenes.barn_rays, 256, 16384, 4)
    print("barn", st, "colour err", err)
