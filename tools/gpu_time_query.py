"""Dev probe: timing of grid build + query at lego scale; prints which HIP runtime got loaded."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointnerf_amd import config, scenes
from pointnerf_amd.point_query import lighting_fast_querier, clear_grid_cache
print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'pnerf' in l][:4])
dev = torch.device('cuda:0')
opt = config.bench_lego_opt()
xyz = torch.from_numpy(scenes.lego_points()).to(dev)
inp = {k: (torch.from_numpy(v) if hasattr(v, 'dtype') else v) for k, v in scenes.random_rays(3, 65536).items()}
qr = lighting_fast_querier(dev, opt)
rd, cp = inp['raydir'].to(dev), inp['campos'].to(dev)
for it in range(3):
    clear_grid_cache(); torch.cuda.synchronize(); t = time.time()
    d = qr.query_dense(xyz[None], xyz.shape[0], 2.0, 6.0, rd, cp); torch.cuda.synchronize()
    print('cold (grid build + query) ms', (time.time() - t) * 1e3)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    d = qr.query_dense(xyz[None], xyz.shape[0], 2.0, 6.0, rd, cp); torch.cuda.synchronize()
    print('warm query ms', (time.time() - t) * 1e3)
print('grid', qr.last_grid_info)
c = d['counters'].cpu().tolist()
print('counters valid_samples, rays_hit, n_sel, n_neigh', c[:4], 'per hit ray: samples %.1f rows %.1f' % (c[0] / max(c[1], 1), c[3] / max(c[1], 1)))
