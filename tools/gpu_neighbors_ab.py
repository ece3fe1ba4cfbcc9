"""dev: time of the query kernels alone at a bench configuration (the library's HIP-event profiler), for A/B runs of library variants"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnerf_amd import ops

cfgname = sys.argv[1] if len(sys.argv) > 1 else "lego"
dev = torch.device("cuda:0")
label, make_opt, points_fn, n_points, rays_fn = bench._cfg()[cfgname]
opt = make_opt(is_train=0)
model = bench.build_model(opt, n_points, dev, points_fn)
inp = bench.step_inputs(0, 0, 1, 65536, dev, rays_fn)
q = model.neural_points.querier
ops.prof_enable(True)
with torch.no_grad():
    for _ in range(3):
        model(**inp)
    ops.prof_collect()
    for _ in range(10):
        model(**inp)
    prof = ops.prof_collect()
st = model.last_stats
print(json.dumps(dict(config=cfgname, neighbors_ms=prof["neighbors"][0] / prof["neighbors"][1], probe_ms=prof["probe"][0] / prof["probe"][1],
                      n_selected=st.get("n_selected"), n_rows=st.get("n_neighbor_rows"), n_valid=st.get("n_valid_samples"))))
