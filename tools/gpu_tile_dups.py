"""dev: how many DISTINCT points does a 64-row aggregator tile touch?  (bounds what de-duplicating the backward's embedding-gradient
atomics inside a tile could save.)  Bench scene, one batch; tiles = 8 consecutive valid samples x K = 8 slots, as the kernels form them."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pointnerf_amd import config

dev = torch.device("cuda:0")
opt = config.bench_lego_opt()
model = bench.build_model(opt, 2_000_000, dev)
inp = bench.step_inputs(0, 0, 1, 65536, dev)
with torch.no_grad():
    dense = model.neural_points.query_dense(inp)
n = int(dense["counters"][0].item())
vl = dense["valid_list"][:n].long()
pidx = dense["sample_pidx"].reshape(-1, opt.K)[vl]                 # [n, K] point ids (-1 = empty slot)
full = (pidx >= 0).all(1)
p = pidx[full][: (int(full.sum()) // 8) * 8].reshape(-1, 64).cpu().numpy()
uniq = np.array([len(np.unique(r)) for r in p[:20000]])
out = {"valid_samples": n, "full_class_samples": int(full.sum()), "tiles_sampled": len(uniq), "distinct_points_per_64_rows_mean": float(uniq.mean()),
       "p10": float(np.percentile(uniq, 10)), "p90": float(np.percentile(uniq, 90))}
print(json.dumps(out))
