"""dev: render step-0 of the bench workload (inference and one training step) and dump the results, to compare two builds of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pointnerf_amd import config, dist as pdist
dev = torch.device("cuda:0")
opt = config.bench_lego_opt(is_train=0)
model = bench.build_model(opt, 2_000_000, dev)
inp = bench.step_inputs(0, 0, 1, 65536, dev)
with torch.no_grad():
    out = model(**inp)
res = {"ray_color": out["coarse_raycolor"].cpu().numpy(), "opacity": out["coarse_point_opacity"].cpu().numpy()}
opt.is_train = 1; opt.ray_jitter = 0.0
npnt = model.neural_points
out = model(**inp)
loss = pdist.hot_path_loss(opt, out, inp["gt_image"])
loss.backward()
res["loss"] = np.float64(loss.item())
res["g_emb"] = npnt.points_embeding.grad.cpu().numpy()
res["g_w1"] = next(p for n, p in model.aggregator.named_parameters() if n == "block1.0.weight").grad.cpu().numpy()
np.savez(sys.argv[1], **res)
print("dumped", sys.argv[1], float(res["loss"]))
