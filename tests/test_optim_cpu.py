"""Host logic of the parameter update without a GPU: FusedAdam's bookkeeping and state layout against
torch.optim.Adam, and the ZeRO-1 ShardedAdam on two gloo ranks against single-process Adam on the summed gradients.
The arithmetic itself is injected here as a torch restatement (test infrastructure); the HIP kernel is checked against
torch.optim.Adam in tests/test_gpu_optim.py."""
import math
import os
import socket

import torch
import torch.multiprocessing as mp

from pointnerf_amd.optim import FusedAdam, ShardedAdam


def adam_restatement(p, g, m, v, lr, b1, b2, eps, step):
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(37, 5, generator=g).requires_grad_(True), torch.randn(256, generator=g).requires_grad_(True), torch.randn(3, 3, 3, generator=g).requires_grad_(True)]


def test_fused_adam_bookkeeping_matches_torch_adam():
    a, b = _params(0), _params(0)
    oa, ob = torch.optim.Adam(a, lr=2e-3, betas=(0.9, 0.999)), FusedAdam(b, lr=2e-3, betas=(0.9, 0.999), update=adam_restatement)
    for it in range(5):
        gen = torch.Generator().manual_seed(100 + it)
        grads = [torch.randn(p.shape, generator=gen) for p in a]
        for p, q, g in zip(a, b, grads):
            p.grad, q.grad = g.clone(), g.clone()
        if it == 3:
            a[2].grad = None; b[2].grad = None               # a parameter without a gradient is skipped, its step does not advance
        oa.step(); ob.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=0, atol=1e-6)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["lr"] == sb["param_groups"][0]["lr"]
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
        assert torch.allclose(sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"], atol=1e-7)
        assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], atol=1e-7)
    # a torch.optim.Adam checkpoint resumes in FusedAdam and the other way round
    c = _params(0)
    oc = FusedAdam(c, lr=1e-3, update=adam_restatement)
    oc.load_state_dict(sa)
    assert oc.param_groups[0]["lr"] == 2e-3 and float(oc.state[c[0]]["step"]) == 5.0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    ps = _params(1)
    opt = ShardedAdam(ps, lr=3e-3, update=adam_restatement)
    for it in range(4):
        for i, p in enumerate(ps):
            gen = torch.Generator().manual_seed(1000 * rank + 10 * it + i)
            p.grad = torch.randn(p.shape, generator=gen)
        opt.step()
        opt.zero_grad()
    if rank == 0:
        torch.save([p.detach() for p in ps], out)
    torch.distributed.destroy_process_group()


def test_sharded_adam_two_ranks_equals_adam_on_summed_gradients(tmp_path):
    world, out = 2, str(tmp_path / "p.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    ref = _params(1)
    opt = torch.optim.Adam(ref, lr=3e-3)
    for it in range(4):
        for i, p in enumerate(ref):
            p.grad = sum(torch.randn(p.shape, generator=torch.Generator().manual_seed(1000 * r + 10 * it + i)) for r in range(world))
        opt.step()
    for p, q in zip(ref, got):
        assert torch.allclose(p.detach(), q, rtol=0, atol=2e-6)
