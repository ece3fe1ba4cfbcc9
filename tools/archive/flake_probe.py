"""dev helper (not a test): is the fused training forward bit-reproducible?  Runs the configs[1] subsample of test_gpu_bench_config N times
in one process and compares every output with the first run's, bit for bit; prints where they differ."""
import sys
import torch
from gpu_util import hip_render
from test_gpu_bench_config import _bench_case

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
train = (sys.argv[2] != "infer") if len(sys.argv) > 2 else True
opt, xyz, attrs, inp, mlp = _bench_case()
if len(sys.argv) > 3 and sys.argv[3] == "poison":            # what tests/conftest.py does to the saved-activation arena
    from pointnerf_amd import ops
    orig = ops.Arena.take
    def take(self, nbytes, device):
        t = orig(self, nbytes, device)
        t.fill_(0xFF)
        return t
    ops.Arena.take = take
first = None
mode = sys.argv[3] if len(sys.argv) > 3 else ""
dummy = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0") if mode == "dummyfill" else None
for it in range(N):
    if dummy is not None:
        dummy.fill_(0xFF)                                   # the timing of the poison fill without its content
    dense, fwd, ctx = hip_render(opt, xyz, attrs, inp, mlp, train=train)
    cur = {k: fwd[k].detach().cpu().clone() for k in ("decoded", "weight", "ray_color", "opacity")}
    cur["pidx"] = dense["sample_pidx"].cpu().clone()
    if first is None:
        first = cur
        continue
    for k in cur:
        if not torch.equal(cur[k], first[k]):
            d = (cur[k].float() - first[k].float()).abs()
            idx = torch.nonzero(d.reshape(-1, d.shape[-1]).amax(-1) > 0)[:, 0]
            print("run %d: %s differs in %d rows, max %.3e; first rows %s" % (it, k, idx.numel(), float(d.max()), idx[:8].tolist()))
            if k == "decoded":
                flat = d.reshape(-1, 4)
                nn = (cur["pidx"].reshape(-1, opt.K) >= 0).sum(-1)
                for i in idx[:8].tolist():
                    print("    sample %d (ray %d, s %d): neighbors %d, |diff| per channel %s" % (i, i // opt.SR, i % opt.SR, int(nn[i]), flat[i].tolist()))
print("done", N, "runs, train =", train, "mode", mode)
