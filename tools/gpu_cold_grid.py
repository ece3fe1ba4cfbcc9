"""dev: what a cold-grid step pays on top of a warm one (bench.py's ms_step_cold_grid): hyper-parameters, grid build, and the isolated step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnerf_amd import config, ops, point_query, dist as pdist
dev = torch.device("cuda:0")
opt = config.bench_lego_opt(is_train=1)
model = bench.build_model(opt, 2_000_000, dev)
npnt = model.neural_points
inp = bench.step_inputs(0, 0, 1, 65536, dev)
def step():
    for p in list(model.aggregator.parameters()) + [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]:
        p.grad = None
    out = model(**inp)
    pdist.hot_path_loss(opt, out, inp["gt_image"]).backward()
for _ in range(3): step()
def timed(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[n // 2]
xyz = npnt.xyz.detach().reshape(-1, 3)
print("warm isolated step           %.2f ms" % timed(step))
def cold():
    point_query.clear_grid_cache(); step()
print("cold isolated step           %.2f ms" % timed(cold))
print("grid_hyperparameters         %.3f ms" % timed(lambda: ops.grid_hyperparameters(opt, xyz)))
r, svs, svd, rad = ops.grid_hyperparameters(opt, xyz)
gp = ops.make_grid_params(r, svs, svd, opt.kernel_size, opt.query_size, opt.P, opt.max_o, rad)
print("make_grid_params (host)      %.3f ms" % timed(lambda: ops.make_grid_params(r, svs, svd, opt.kernel_size, opt.query_size, opt.P, opt.max_o, rad)))
print("build_grid                   %.3f ms" % timed(lambda: ops.build_grid(gp, xyz)))
