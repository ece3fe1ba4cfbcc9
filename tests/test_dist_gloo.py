"""world_size-2 test (gloo, CPU) of the ray-shard data-parallel logic in pointnerf_amd/dist.py: the loss with global
normalisers + summed gradients equals the single-process loss/gradients on the whole batch.  The per-rank compute is
the CPU oracle (test infrastructure); what is under test is the sharding, the loss normalisation and the all-reduce."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cases import build_case
from pointnerf_amd import dist as pdist
from oracle import pyref


def _grads(opt, xyz, attrs, inp, mlp, sl):
    mlp = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    pts = dict(xyz=xyz, **{k: v.clone().requires_grad_(True) for k, v in attrs.items()})
    sub = dict(inp)
    sub["raydir"], sub["gt_image"] = inp["raydir"][:, sl].contiguous(), inp["gt_image"][:, sl].contiguous()
    out = pyref.render(opt, pts, mlp, sub)
    loss = pdist.hot_path_loss(opt, out, sub["gt_image"])
    loss.backward()
    mp_, pp_ = list(mlp.values()), [pts[k] for k in attrs]
    for p in mp_ + pp_:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    return loss.detach(), mp_, pp_


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    opt, xyz, attrs, inp, mlp = build_case("small_k4")
    R = inp["raydir"].shape[1]
    loss, mp_, pp_ = _grads(opt, xyz, attrs, inp, mlp, pdist.shard_slice(R))
    pdist.allreduce_grads(mp_, pp_)
    dist.all_reduce(loss)
    if rank == 0:
        q.put((loss.item(), [p.grad.numpy() for p in mp_], [p.grad.numpy() for p in pp_]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ray_shard_equals_single_process():
    torch.set_num_threads(2)
    opt, xyz, attrs, inp, mlp = build_case("small_k4")
    loss1, m1, p1 = _grads(opt, xyz, attrs, inp, mlp, slice(None))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss2, m2, p2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(loss2 - loss1.item()) <= 1e-5 * abs(loss1.item())
    for a, b in zip(m1, m2):
        assert np.abs(a.grad.numpy() - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-8) + 1e-9
    for a, b in zip(p1, p2):      # point 0 sums thousands of empty-slot terms: order-of-summation noise
        assert np.abs(a.grad.numpy() - b).max() <= 5e-4 * max(np.abs(b).max(), 1e-8) + 1e-9
