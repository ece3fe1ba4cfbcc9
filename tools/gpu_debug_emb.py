import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from pointnerf_amd import config, scenes, ops
from oracle import pyref
import test_gpu_backward as T
opt = config.chair_opt()
xyz = torch.from_numpy(scenes.chair_points())
attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(8192, 32, 0).items()}
inp = pyref.to_torch_inputs(scenes.block_rays())
mlp = pyref.init_mlp_params(opt, seed=0, bias_scale=0.05)
gm_o, gp_o, probe = T._oracle_grads(opt, xyz, attrs, inp, mlp)
gm, gp, fwd, hit = T._hip_grads(opt, xyz, attrs, inp, mlp, probe)
a, b = gp['points_embeding'], gp_o['points_embeding']
err = (a - b).abs()
print('max err', err.max().item(), 'mean err', err.mean().item(), 'mean |b|', b.abs().mean().item())
flat = err.flatten().topk(12)
for v, i in zip(flat.values.tolist(), flat.indices.tolist()):
    p, d = divmod(i, 32)
    print('pt %5d dim %2d ours % .6e ref % .6e err %.2e emb % .4f' % (p, d, a[p, d].item(), b[p, d].item(), v, attrs['points_embeding'][0, p, d].item()))
print('frac elems with err > 1e-6:', (err > 1e-6).float().mean().item(), ' > 1e-5:', (err > 1e-5).float().mean().item())
perdim = err.max(0).values
print('per-dim max err', ['%.1e' % x for x in perdim.tolist()])
# second run to see run-to-run (atomic order) noise
gm2, gp2, _, _ = T._hip_grads(opt, xyz, attrs, inp, mlp, probe)
print('run-to-run diff', (gp2['points_embeding'] - a).abs().max().item())
