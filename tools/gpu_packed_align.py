"""dev: do k_agg_forward2 / k_agg_backward depend on WHERE the 2.7 MB packed weight image sits (all 256 CUs stream it from L2
continuously)?  The image is placed at different offsets inside a big buffer and the kernels are timed."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnerf_amd import config, ops, dist as pdist, _lib

dev = torch.device("cuda:0")
opt = config.bench_lego_opt(is_train=1)
model = bench.build_model(opt, 2_000_000, dev)
agg, npnt = model.aggregator, model.neural_points
params = [p for p in agg.parameters()] + [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
inputs = [bench.step_inputs(i, 0, 1, 65536, dev) for i in range(3)]
nbytes = _lib.lib().pnerf_mlp_packed_bytes()
big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
base = (-big.data_ptr()) % (4 << 20)
st = agg.mlp_state() if hasattr(agg, "mlp_state") else None
offs = [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 256 * 7, 3 << 20, 5 << 20, (8 << 20) + 128, 16 << 20]
if len(sys.argv) > 1:
    offs = [int(x) for x in sys.argv[1:]]
ops.prof_enable(True)
for off in offs:
    st.packed = big[base + off: base + off + nbytes]
    for i in range(3):
        for p in params:
            p.grad = None
        out = model(**inputs[i])
        loss = pdist.hot_path_loss(opt, out, inputs[i]["gt_image"])
        loss.backward()
        if i == 0:
            torch.cuda.synchronize(); ops.prof_collect()
    torch.cuda.synchronize()
    prof = ops.prof_collect()
    print(json.dumps({"offset": off, "ptr": hex(st.packed.data_ptr()), "agg_forward": round(prof["agg_forward"][0] / 2, 2), "agg_backward": round(prof["agg_backward"][0] / 2, 2),
                      "wgrad": round(prof["wgrad"][0] / 2, 2)}), flush=True)
