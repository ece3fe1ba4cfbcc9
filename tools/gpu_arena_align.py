"""dev: how much do k_agg_forward / k_agg_backward / k_wgrad depend on WHERE the saved-activation arena sits in the address space?
(found by accident: enlarging an unrelated workspace by 11 MB made k_agg_backward 33 % slower with a bit-identical kernel)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnerf_amd import config, ops, dist as pdist

dev = torch.device("cuda:0")
opt = config.bench_lego_opt(is_train=1)
model = bench.build_model(opt, 2_000_000, dev)
agg, npnt = model.aggregator, model.neural_points
params = [p for p in agg.parameters()] + [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
inputs = [bench.step_inputs(i, 0, 1, 65536, dev) for i in range(3)]
MB = 1 << 20
configs = [(0, 0), (0, 12058624), (0, 4 * MB), (0, 32 * MB), (0, 1 * MB), (0, 12058624 + 2 * MB), (0, 0)]     # (arena align, workspace pad)
if len(sys.argv) > 1:
    configs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
ops.DEBUG_PTRS = {}
ops.prof_enable(True)
for align, off in configs:
    ops.ARENA.free.clear(); ops.ARENA.align, ops.WS_PAD = align, off
    torch.cuda.empty_cache()
    base = None
    for i in range(3):
        for p in params:
            p.grad = None
        out = model(**inputs[i])
        loss = pdist.hot_path_loss(opt, out, inputs[i]["gt_image"])
        loss.backward()
        if i == 0:
            torch.cuda.synchronize(); ops.prof_collect()
            base = ops.ARENA.free[0].data_ptr() if ops.ARENA.free else None
    torch.cuda.synchronize()
    prof = ops.prof_collect()
    ms = {k: round(prof[k][0] / prof[k][1], 2) for k in ("agg_forward", "agg_backward", "wgrad")}
    ms["wgrad"] = round(prof["wgrad"][0] / 2, 2)
    print(json.dumps({"align": align, "ws_pad": off, **ms, "ptrs": {k: hex(v) for k, v in ops.DEBUG_PTRS.items()}}), flush=True)
