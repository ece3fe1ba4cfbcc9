import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cases, gpu_util
opt, xyz, attrs, inp, mlp = cases.build_case("small_k8")
outs = {}
for train in (False, True):
    dense, fwd, ctx = gpu_util.hip_render(opt, xyz, attrs, inp, mlp, train=train)
    torch.cuda.synchronize()
    outs[train] = fwd["decoded"].cpu().reshape(-1, 4)
    vl = dense["valid_list"][:ctx["n_valid"]].cpu().numpy()
d = (outs[True] - outs[False]).abs().max(dim=1)[0].numpy()
bad_si = np.nonzero(d > 1e-5)[0]
pos = {int(s): i for i, s in enumerate(vl)}
vs = np.array(sorted(pos[int(s)] for s in bad_si if int(s) in pos))
print("n_valid", len(vl), "bad samples", len(bad_si), "in list", len(vs))
tiles = vs // 8
print("bad tiles:", np.unique(tiles)[:40], "...", "count per tile:", np.bincount(tiles)[np.unique(tiles)][:40])
print("within-tile positions of bad:", np.bincount(vs % 8, minlength=8))
print("colour tiles (64):", np.unique(vs // 64))
