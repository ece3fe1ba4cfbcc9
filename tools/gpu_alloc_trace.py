"""dev: which hipMalloc calls (new caching-allocator segments) happen in steady-state training steps of the bench workload?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnerf_amd import config, dist as pdist
from pointnerf_amd.optim import FusedAdam

dev = torch.device("cuda:0")
opt = config.bench_lego_opt(is_train=1)
model = bench.build_model(opt, 2_000_000, dev)
agg, npnt = model.aggregator, model.neural_points
mlp = [p for p in agg.parameters()]
pts = [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
o1, o2 = FusedAdam(mlp, lr=opt.lr), FusedAdam(pts, lr=opt.plr)
seen = set()
for i in range(8):
    inp = bench.step_inputs(i, 0, 1, 65536, dev)
    torch.cuda.synchronize()
    n0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    o1.zero_grad(set_to_none=True); o2.zero_grad(set_to_none=True)
    out = model(**inp)
    loss = pdist.hot_path_loss(opt, out, inp["gt_image"])
    loss.backward()
    o1.step(); o2.step()
    torch.cuda.synchronize()
    snap = torch.cuda.memory_snapshot()
    new = [(s["total_size"], s["segment_type"]) for s in snap if s["address"] not in seen]
    seen = {s["address"] for s in snap}
    st = torch.cuda.memory_stats(dev)
    print(json.dumps({"step": i, "n_hit": model.last_stats["rays_hit"], "n_valid": model.last_stats["n_valid_samples"],
                      "device_allocs": st.get("num_device_alloc", 0) - n0, "device_frees_total": st.get("num_device_free", 0),
                      "new_segments_MB": sorted([round(a / 2 ** 20, 1) for a, _ in new], reverse=True)[:12],
                      "reserved_GB": round(st["reserved_bytes.all.current"] / 2 ** 30, 2), "active_GB": round(st["active_bytes.all.current"] / 2 ** 30, 2),
                      "active_peak_GB": round(st["active_bytes.all.peak"] / 2 ** 30, 2)}))
