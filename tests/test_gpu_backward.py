"""GPU parity of the HIP backward (ray-march, colour MLP, aggregator MLP dgrad, gather scatter-add, weight-grad
GEMMs) against torch.autograd over the CPU oracle.  Tolerances: see _check (5e-4 of each MLP tensor's max |grad|, 2e-4 / 5e-3
for the per-point tensors); the oracle's own gradients are pinned to the reference modules' gradients by
tests/test_oracle_golden.py."""
import numpy as np
import pytest
import torch

from cases import CASES, build_case
from gpu_util import hip_render, DEV
from pointnerf_amd import config, scenes, ops
from oracle import pyref

pytestmark = pytest.mark.gpu


def _hip_grads(opt, xyz, attrs, inp, mlp, probe_hit):
    dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp, train=True)
    dev = torch.device(DEV)
    hit = dense["ray_hit"] > 0
    g = torch.zeros(ctx["R"], 3, device=dev)
    g[hit] = probe_hit.to(dev)
    gflat = torch.zeros_like(ctx["flat"])
    grads = {k: torch.zeros_like(v) for k, v in ctx["pts_t"].items()}
    ops.render_backward(ctx["cam"], ctx["pts"], ctx["packed"], ctx["flat"], ctx["raydir"], dense, ctx["R"], opt.SR, opt.K,
                        ctx["n_valid"], fwd, g, gflat, grads)
    torch.cuda.synchronize()
    lay, _ = ops.mlp_layout()
    gm = {k: gflat[o:o + int(np.prod(shp))].view(shp).cpu() for k, (o, shp) in lay.items()}
    return gm, {k: v.cpu() for k, v in grads.items()}, fwd, hit.cpu()


def _oracle_grads(opt, xyz, attrs, inp, mlp, probe=None):
    torch.set_num_threads(8)
    mlp = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    points = dict(xyz=xyz, **{k: v.clone().requires_grad_(True) for k, v in attrs.items()})
    out = pyref.render(opt, points, mlp, inp, nthreads=8)
    col = out["coarse_raycolor"]
    if probe is None:
        probe = torch.rand(col.shape, generator=torch.Generator().manual_seed(123))
    (col * probe).sum().backward()
    return {k: v.grad for k, v in mlp.items()}, {k: points[k].grad[0] for k in attrs}, probe[0]


def _check(name, a, b, point_cap=5e-3):
    """MLP tensors: every element within 5e-4 of the tensor's max |grad| (measured on MI355X over all cases of this file: worst 1.95e-4, on
    block1.0 of a 12-ray case; typical 1e-5 -- fp32 accumulation order on our side and on the oracle's, plus the one-plane dY of the
    weight-gradient GEMM, tests/test_split_f16_cpu.py).
    Point tensors: >= 99.8 % of the elements within 2e-4 of max |grad|, and no element beyond 5e-3 (measured worst 1.1e-3).  The slack
    exists because LeakyReLU's derivative is discontinuous: among ~1e8 hidden activations a handful have |pre-activation| < 1e-7, where an
    ulp of summation-order noise flips 1 <-> 0.01 for that unit of that row on one side only; tests/test_gpu_bench_config.py attributes
    every such outlier to a kinked row explicitly at the bench configuration."""
    scale = max(float(b.abs().max()), 1e-8)
    e = (a - b).abs()
    err = float(e.max())
    point = name.startswith("points_")
    tol, cap = (2e-4, point_cap) if point else (5e-4, 5e-4)
    frac_bad = float((e > tol * scale).float().mean())
    print("%-28s max|grad| %.3e  err %.3e  rel %.2e  frac>tol %.1e" % (name, scale, err, err / scale, frac_bad))
    assert frac_bad <= (2e-3 if point else 0.0), (name, frac_bad)
    assert err <= cap * scale, (name, err, scale)


def _run(opt, xyz, attrs, inp, mlp):
    gm_o, gp_o, probe = _oracle_grads(opt, xyz, attrs, inp, mlp)
    gm, gp, fwd, hit = _hip_grads(opt, xyz, attrs, inp, mlp, probe)
    for k in gm_o:
        _check(k, gm[k], gm_o[k])
    for k in gp_o:
        _check(k, gp[k], gp_o[k])


@pytest.mark.parametrize("name", list(CASES))
def test_backward_matches_oracle(name):
    opt, xyz, attrs, inp, mlp = build_case(name)
    _run(opt, xyz, attrs, inp, mlp)


# K % 4 == 0 runs three sample classes (K, K/2, K/4 rows per sample), other K one; a single ray exercises the odd / empty tile
# pairs of the two-tile kernels
@pytest.mark.parametrize("K,SR,size", [(12, 20, 12), (3, 70, 12), (16, 12, 12), (1, 8, 12), (2, 16, 9), (4, 24, 12), (5, 20, 7), (8, 128, 1)])
def test_backward_other_K(K, SR, size):
    opt = config.lego_opt(K=K, SR=SR, P=24, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    xyz = torch.from_numpy(scenes.chair_points(2500, seed=5, radius=0.06))
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(2500, 32, 5).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=55.0, x0=394, y0=394, size=size))
    mlp = pyref.init_mlp_params(opt, seed=3, bias_scale=0.1)
    _run(opt, xyz, attrs, inp, mlp)


def test_backward_config1_chair():
    opt = config.chair_opt()
    xyz = torch.from_numpy(scenes.chair_points())
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(8192, 32, 0).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays())
    mlp = pyref.init_mlp_params(opt, seed=0, bias_scale=0.05)
    _run(opt, xyz, attrs, inp, mlp)


def _model_grads(opt, xyz, attrs, inp, mlp, dev):
    """loss.backward() through NeuralPointsRayMarching (the fused autograd node): gradients of every parameter"""
    from pointnerf_amd.neural_points import NeuralPoints
    from pointnerf_amd.point_aggregators import PointAggregator
    from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
    agg = PointAggregator(opt).to(dev)
    agg.load_state_dict(mlp)
    agg.flatten_()
    npnt = NeuralPoints(32, xyz.shape[0], opt, torch.device(dev))
    a = {k: v.to(dev) for k, v in attrs.items()}
    npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"], parameter=True)
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    out = model(**d)
    pyref.training_loss(opt, out, {"gt_image": d["gt_image"]}).backward()
    g = {n: p.grad.detach().cpu().clone() for n, p in agg.named_parameters()}
    g.update({n: getattr(npnt, n).grad.detach().cpu().clone() for n in ("points_embeding", "points_conf", "points_dir", "points_color")})
    return g


def _chunked_equals_one_pass(opt, xyz, attrs, inp, mlp, dev, budget_gb, monkeypatch):
    from pointnerf_amd.fused import FusedRender
    one = _model_grads(opt, xyz, attrs, inp, mlp, dev)
    monkeypatch.setenv("PNERF_ARENA_BUDGET_GB", str(budget_gb))
    FusedRender.last_chunks = None
    many = _model_grads(opt, xyz, attrs, inp, mlp, dev)
    assert FusedRender.last_chunks is not None and FusedRender.last_chunks[0] < FusedRender.last_chunks[1], "the budget did not force chunks"
    for k in one:
        scale = max(float(one[k].abs().max()), 1e-12)
        assert float((one[k] - many[k]).abs().max()) <= 2e-5 * scale, k
    return FusedRender.last_chunks


def test_backward_by_ray_chunks_equals_one_pass(monkeypatch):
    """a render step whose saved activations exceed the arena budget: forward without saving, backward re-runs the forward per run of
    rays (fused._backward_in_chunks); same gradients as the one-pass step"""
    opt = config.lego_opt(K=8, SR=24, P=12, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    xyz = torch.from_numpy(scenes.chair_points(1500, seed=0, radius=0.06))
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(1500, 32, 0).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=30.0, x0=394, y0=394, size=12))
    mlp = pyref.init_mlp_params(opt, seed=0, bias_scale=0.1)
    step, R = _chunked_equals_one_pass(opt, xyz, attrs, inp, mlp, DEV, 0.02, monkeypatch)
    print("chunks of", step, "rays of", R)


def fused_losses_equivalence(opt, xyz, attrs, inp, mlp, dev):
    """the step's loss and every gradient are the same whichever way the two loss terms are formed: (a) the ATen chains on the compacted
    [1, R'', ...] outputs and a materialised conf_coefficient; (b) the zero-one regulariser as part of the render node (numerator = an output
    of the node, its conf gradient on the node's own conf atomics + the closed form for the empty slots), colour loss on the compacted
    outputs; (c) that and the colour loss over the dense ray colours (ops.ColorLossRays)"""
    from pointnerf_amd import dist as pdist
    from pointnerf_amd.neural_points import NeuralPoints
    from pointnerf_amd.point_aggregators import PointAggregator
    from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}

    def run(fused_zo, fused_col):
        agg = PointAggregator(opt).to(dev)
        agg.load_state_dict(mlp)
        agg.flatten_()
        npnt = NeuralPoints(32, xyz.shape[0], opt, torch.device(dev))
        a = {k: v.to(dev) for k, v in attrs.items()}
        npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"], parameter=True)
        model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
        model.fused_zero_one, model.fused_color_loss = fused_zo, fused_col
        out = model(**d)
        assert ("_dense_color" in out) == fused_col and ("coarse_raycolor" in out) != fused_col
        assert ("_zero_one_sum" in out) == fused_zo and ("conf_coefficient" in out) != fused_zo
        loss = pdist.hot_path_loss(opt, out, d["gt_image"])
        loss.backward()
        g = {n: p.grad.detach().cpu().clone() for n, p in agg.named_parameters()}
        g.update({n: getattr(npnt, n).grad.detach().cpu().clone() for n in ("points_embeding", "points_conf", "points_dir", "points_color")})
        return float(loss.detach()), g

    l0, g0 = run(False, False)
    assert float(g0["points_conf"].abs().max()) > 0
    for mode in ((True, False), (True, True)):
        l1, g1 = run(*mode)
        assert abs(l0 - l1) <= 2e-6 * abs(l0), (mode, l0, l1)
        for k in g0:            # (the backward's atomics reorder between two runs on the device: 1e-5 of a tensor's largest element)
            assert torch.allclose(g0[k], g1[k], rtol=1e-4, atol=1e-5 * float(g0[k].abs().max())), (mode, k, float((g0[k] - g1[k]).abs().max()), float(g0[k].abs().max()))


def test_fused_losses_match_the_aten_chains():
    fused_losses_equivalence(*build_case("small_k8"), DEV)
