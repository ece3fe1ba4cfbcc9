"""End to end: three optimisation steps of the hot path on the device (NeuralPointsRayMarching forward, the reference's
losses, backward through the HIP kernels, one-pass HIP Adam on the MLP and on the point parameters) against the same three
steps of the CPU oracle with torch.optim.Adam -- the reference's loop body (models/mvs_points_volumetric_model.py:98-118,
base_rendering_model.py:533-662) as a whole, not kernel by kernel.  The loss trajectory must agree to 1e-4 relative, the
parameters after three steps to the tolerance of the backward test."""
import pytest
import torch

from cases import build_case
from pointnerf_amd import dist as pdist
from pointnerf_amd.neural_points import NeuralPoints
from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
from pointnerf_amd.optim import FusedAdam
from pointnerf_amd.point_aggregators import PointAggregator
from oracle import pyref

pytestmark = pytest.mark.gpu
STEPS = 3


def oracle_steps(opt, xyz, attrs, inp, mlp, steps):
    """`steps` optimisation steps of the CPU oracle with torch.optim.Adam: (losses, MLP tensors, point tensors)"""
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    oa = {k: v.clone().requires_grad_(True) for k, v in attrs.items()}
    o_mlp = torch.optim.Adam(list(om.values()), lr=opt.lr, betas=(0.9, 0.999))
    o_pts = torch.optim.Adam(list(oa.values()), lr=opt.plr, betas=(0.9, 0.999))
    ref_losses = []
    for _ in range(steps):
        o_mlp.zero_grad(); o_pts.zero_grad()
        out = pyref.render(opt, dict(xyz=xyz, **oa), om, inp, nthreads=8)
        loss = pyref.training_loss(opt, out, inp)
        loss.backward()
        o_mlp.step(); o_pts.step()
        ref_losses.append(float(loss))
    return ref_losses, om, oa


def device_steps(opt, xyz, attrs, inp, mlp, steps):
    """the same steps on the device: (losses, aggregator, neural points)"""
    dev = torch.device("cuda:0")
    agg = PointAggregator(opt).to(dev)
    agg.load_state_dict(mlp)
    agg.flatten_()
    npnt = NeuralPoints(32, xyz.shape[0], opt, dev)
    a = {k: v.to(dev) for k, v in attrs.items()}
    npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"],
                    points_conf=a["points_conf"], parameter=True)
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt).to(dev)
    agg.flatten_()
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    mlp_params = [p for p in agg.parameters() if p.requires_grad]
    pt_params = [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
    h_mlp, h_pts = FusedAdam(mlp_params, lr=opt.lr, betas=(0.9, 0.999)), FusedAdam(pt_params, lr=opt.plr, betas=(0.9, 0.999))
    losses = []
    for _ in range(steps):
        h_mlp.zero_grad(set_to_none=True); h_pts.zero_grad(set_to_none=True)
        out = model(**d)
        loss = pdist.hot_path_loss(opt, out, d["gt_image"])
        loss.backward()
        h_mlp.step(); h_pts.step()
        losses.append(loss.detach())
    return [float(v) for v in torch.stack(losses).cpu()], agg, npnt


@pytest.mark.parametrize("name", ["small_k8", "small_k4"])
def test_three_training_steps_match_the_oracle(name):
    opt, xyz, attrs, inp, mlp = build_case(name)
    ref_losses, om, oa = oracle_steps(opt, xyz, attrs, inp, mlp, STEPS)
    losses, agg, npnt = device_steps(opt, xyz, attrs, inp, mlp, STEPS)
    print("losses  device", losses, " oracle", ref_losses)
    for a_, b_ in zip(losses, ref_losses):
        assert abs(a_ - b_) <= 1e-4 * max(1.0, abs(b_)), (losses, ref_losses)
    # parameters after STEPS updates: Adam normalises the step to ~lr per element, so a LeakyReLU-kink flip in a gradient of
    # magnitude ~0 can move one element by up to 2*lr per step; everything else agrees to rounding
    sd = agg.state_dict()
    for k, v in om.items():
        e = (sd[k].detach().cpu() - v.detach()).abs()
        frac = float((e > 0.05 * opt.lr).float().mean())
        print("%-26s max err %.2e  frac > 5%% of lr: %.1e" % (k, float(e.max()), frac))
        assert float(e.max()) <= 2.0 * opt.lr * STEPS and frac <= 2e-3, (k, float(e.max()), frac)
    for k, t in (("points_embeding", npnt.points_embeding), ("points_conf", npnt.points_conf), ("points_dir", npnt.points_dir), ("points_color", npnt.points_color)):
        e = (t.detach().cpu().reshape(oa[k].shape) - oa[k].detach()).abs()
        frac = float((e > 0.05 * opt.plr).float().mean())
        print("%-26s max err %.2e  frac > 5%% of plr: %.1e" % (k, float(e.max()), frac))
        assert float(e.max()) <= 2.0 * opt.plr * STEPS and frac <= 2e-3, (k, float(e.max()), frac)


def _speculative_runs(batches_of):
    """the batches of ``batches_of(d)`` with PNERF_SPECULATE off and on: ((loss, gradients) per step, (enqueued ahead, dropped, valid samples) per step, free arena blocks)"""
    from pointnerf_amd import ops, neural_points_volumetric_model as NM
    from pointnerf_amd import dist as pdist
    from pointnerf_amd.neural_points import NeuralPoints
    from pointnerf_amd.point_aggregators import PointAggregator
    opt, xyz, attrs, inp, mlp = build_case("small_k8")
    dev = torch.device("cuda:0")
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    R = d["raydir"].shape[1]
    first = lambda n: dict(d, raydir=d["raydir"][:, :n].contiguous(), gt_image=d["gt_image"][:, :n].contiguous(), pixel_idx=d["pixel_idx"][:, :n].contiguous())
    batches = batches_of(first, R)

    def run(speculate):
        NM.SPECULATE = speculate
        ops.ARENA.free = []
        agg = PointAggregator(opt).to(dev)
        agg.load_state_dict(mlp)
        agg.flatten_()
        npnt = NeuralPoints(32, xyz.shape[0], opt, dev)
        a = {k: v.to(dev) for k, v in attrs.items()}
        npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"], parameter=True)
        model = NM.NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
        model.fused_zero_one = model.fused_color_loss = True
        params = list(agg.parameters()) + [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
        res, flags, drops, nvalid = [], [], [], []
        for batch in batches:
            for p in params:
                p.grad = None
            out = model(**batch)
            flags.append(model.last_stats["enqueued_before_host_read"])
            drops.append(model.last_stats["speculative_result_dropped"])
            nvalid.append(model.last_stats["n_valid_samples"])
            loss = pdist.hot_path_loss(opt, out, batch["gt_image"])
            loss.backward()
            res.append((float(loss.detach()), [p.grad.detach().cpu().clone() for p in params]))
        return res, (flags, drops, nvalid), len(ops.ARENA.free)

    try:
        ref, f0, _ = run(False)
        got, f1, nfree = run(True)
    finally:
        NM.SPECULATE = True
    for (l0, g0), (l1, g1) in zip(ref, got):
        assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
        for a_, b_ in zip(g0, g1):
            assert torch.allclose(a_, b_, rtol=1e-4, atol=2e-5 * float(a_.abs().max()))
    return f0, f1, nfree


def test_step_enqueued_before_its_host_read_and_the_redo_when_the_arena_is_too_small():
    """A training step is enqueued with the arena's capacity as its bound before its counters reach the host (render_dense): (i) a step whose
    count exceeds the capacity (a batch larger than every one before it) drops the speculative result and runs again -- same loss and
    gradients as with PNERF_SPECULATE=0; (ii) the next step of that size IS enqueued ahead; (iii) the arena does not leak."""
    f0, f1, nfree = _speculative_runs(lambda first, R: (first(R // 8), first(R), first(R)))   # small batch sizes the arena; the big one exceeds it; the third fits
    assert f0[0] == [False, False, False] and f1[0] == [False, False, True], (f0, f1)      # first: no arena yet; second: redo; third: ahead
    assert f1[1] == [False, True, False] and f1[2][0] > 0 and f1[2][1] > 2 * f1[2][0], f1          # the second step WAS enqueued ahead and dropped
    assert nfree == 1, nfree


def test_two_consecutive_steps_that_exceed_the_arena():
    """a batch larger than the arena, then a larger one still: both speculative results are dropped and redone (each redo grows the arena; the too-small
    block is released, not kept), a smaller batch in between is enqueued ahead, and every step's loss and gradients equal PNERF_SPECULATE=0."""
    f0, f1, nfree = _speculative_runs(lambda first, R: (first(R // 8), first(R // 2), first(R), first(R // 2), first(R)))
    nv = f1[2]
    assert nv[0] > 0 and nv[1] > 1.3 * nv[0] and nv[2] > 1.3 * nv[1], nv             # (the arena's headroom is 15 %)
    assert f0[0] == [False] * 5 and f0[1] == [False] * 5, f0
    assert f1[1] == [False, True, True, False, False], f1                               # dropped: the two that outgrew the arena
    assert f1[0] == [False, False, False, True, True], f1                               # kept as enqueued: the two that fit
    assert nfree == 1, nfree


def test_speculative_step_with_two_ranks_whose_counts_differ():
    """world 2 (gloo, both ranks on cuda:0): in the second step rank 0 redoes, rank 1 does not; the collectives still pair up, the replicas agree and
    the result equals PNERF_SPECULATE=0 (tests/spec_two_ranks.py)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    import socket
    with socket.socket() as sk:                       # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "spec_two_ranks.py")]
    out = subprocess.check_output(cmd, cwd=os.path.dirname(here), env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600, stderr=subprocess.STDOUT).decode()
    recs = json.loads([l for l in out.splitlines() if l.startswith("[{")][-1])
    r0, r1 = recs
    assert r0["ahead_without"] == [False] * 3 and r1["ahead_without"] == [False] * 3
    assert r0["dropped"] == [False, True, False] and r0["ahead"] == [False, False, True], r0          # rank 0: redo in step 2
    assert r1["dropped"] == [False, False, False] and r1["ahead"] == [False, True, True], r1          # rank 1: enqueued ahead in step 2, kept
    for r in recs:
        assert r["worst_loss_rel"] <= 1e-6 and r["worst_grad_rel"] <= 1e-4 and r["replica_spread"] == 0.0, r
