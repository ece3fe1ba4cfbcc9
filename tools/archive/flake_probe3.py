"""dev helper (not a test): run-to-run spread of the GRADIENTS (forward in training mode + backward of the configs[1] subsample, repeated on the
same inputs).  The backward adds with atomics, so bits differ; what must not happen is a deviation beyond summation-order noise."""
import sys
import numpy as np
import torch
from gpu_util import hip_render, DEV
from pointnerf_amd import ops
from test_gpu_bench_config import _bench_case

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
opt, xyz, attrs, inp, mlp = _bench_case()
dev = torch.device(DEV)
probe = torch.rand(1, 768, 3, generator=torch.Generator().manual_seed(123))
first, worst = None, {}
for it in range(N):
    dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp, train=True)
    hit = dense["ray_hit"] > 0
    g = torch.zeros(ctx["R"], 3, device=dev)
    g[hit] = probe[0, : int(hit.sum())].to(dev)
    gflat = torch.zeros_like(ctx["flat"])
    grads = {k: torch.zeros_like(v) for k, v in ctx["pts_t"].items()}
    ops.render_backward(ctx["cam"], ctx["pts"], ctx["packed"], ctx["flat"], ctx["raydir"], dense, ctx["R"], opt.SR, opt.K, ctx["n_valid"], fwd, g, gflat, grads)
    torch.cuda.synchronize()
    ops.ARENA.give(fwd["saved"])
    cur = {"mlp": gflat.cpu(), **{k: v.cpu() for k, v in grads.items()}}
    if first is None:
        first = cur
        continue
    for k, v in cur.items():
        scale = float(first[k].abs().max())
        dev_k = float((v - first[k]).abs().max()) / max(scale, 1e-20)
        if dev_k > worst.get(k, (0.0, -1))[0]:
            worst[k] = (dev_k, it)
        if dev_k > 1e-4:
            idx = int((v - first[k]).abs().reshape(-1).argmax())
            print("run %d: %s deviates by %.3e of its max (element %d: %.6e vs %.6e)" % (it, k, dev_k, idx, float(v.reshape(-1)[idx]), float(first[k].reshape(-1)[idx])))
print("done", N, "runs; worst run-to-run deviation / max |grad| per tensor:", {k: "%.2e (run %d)" % v for k, v in worst.items()})
