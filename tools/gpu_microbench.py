"""Roofline numbers of the HBM-bound kernels of the path, measured in isolation with the library's HIP-event profiler:
the per-neighbor gather (pnerf_gather_rows, the stand-alone form of NeuralPoints.forward's index_select), the ray probe,
the neighbor query and the ray-march, at BASELINE configs[1] size.  Prints one JSON object (goes to profiles/)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnerf_amd import config, ops, scenes

dev = torch.device("cuda:0")
opt = config.bench_lego_opt(is_train=0)
model = bench.build_model(opt, 2_000_000, dev)
npnt = model.neural_points
inp = bench.step_inputs(0, 0, 1, 65536, dev)
res = {}
ops.prof_enable(True)
with torch.no_grad():
    for _ in range(3):
        out = model(**inp)
    ops.prof_collect()
    for _ in range(10):
        out = model(**inp)
    st = model.last_stats
    prof = ops.prof_collect()
    dense = npnt.querier.last_dense
    R, SR, K, D = 65536, opt.SR, opt.K, opt.z_depth_dim
    nsel, nrows, nval = st["n_selected"], st["n_neighbor_rows"], st["n_valid_samples"]
    ms = lambda k: prof[k][0] / prof[k][1]
    # algorithmic bytes
    res["probe"] = dict(ms=ms("probe"), bytes=R * 24 + R * D / 8 + R * SR * 12 + R * 4)
    res["neighbors"] = dict(ms=ms("neighbors"), bytes=nsel * 27 * 8 + nrows * 16 * 2 + R * SR * (12 + 4 * K + 4))
    res["raymarch_forward"] = dict(ms=ms("raymarch_forward"), bytes=R * SR * (16 + 12 + 4 + 4 + 4) + R * 16)
    # stand-alone gather of the 32-float embedding rows for every neighbor slot of the hit rays
    hit = dense["ray_hit"] > 0
    pidx = dense["sample_pidx"][hit].contiguous()
    emb = npnt.points_embeding.detach().reshape(-1, 32)
    for _ in range(3):
        g = ops.gather_rows(emb, pidx)
    ops.prof_collect()
    for _ in range(10):
        g = ops.gather_rows(emb, pidx)
    p2 = ops.prof_collect()
    n = pidx.numel()
    valid = int((pidx >= 0).sum())
    res["gather_rows_emb32"] = dict(ms=p2["gather"][0] / p2["gather"][1], bytes=n * (4 + 128 + 128), slots=n, valid_slots=valid)
# full-image evaluation: one 800 x 800 view through eval_loop.render_image (4 chunks, grid cached, canvas on the device)
import time
from pointnerf_amd import eval_loop
d0 = scenes.random_rays(0, 16)
cam = {k: torch.from_numpy(__import__("numpy").ascontiguousarray(d0[k])).to(dev) for k in ("campos", "camrotc2w", "near", "far", "bg_color")}
intr = torch.from_numpy(__import__("numpy").asarray(scenes.synth_camera(0.0)[1], dtype="float32")).to(dev)
with torch.no_grad():
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, hit = eval_loop.render_image(model, cam["campos"], cam["camrotc2w"], intr, 800, 800, cam["near"], cam["far"], cam["bg_color"])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
full = dict(ms=dt * 1e3, rays=640000, rays_per_s=640000 / dt, rays_hit=int(hit.sum()))
with torch.no_grad():        # the two-product inference option (ops.set_inference_products(2): weights' high plane only)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img2, hit2 = eval_loop.render_image(model, cam["campos"], cam["camrotc2w"], intr, 800, 800, cam["near"], cam["far"], cam["bg_color"], products=2)
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
full["two_products"] = dict(ms=dt2 * 1e3, rays_per_s=640000 / dt2, max_abs_colour_difference=float((img2 - img).abs().max()))
# point initialisation: voxel down-sampling of a raw cloud, lego script resolution
from pointnerf_amd import point_init
raw = torch.from_numpy(scenes.lego_points(2_000_000)).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cen, gidx, midx = point_init.construct_vox_points_closest(raw, 320)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
vox = dict(ms=dt * 1e3, points=int(raw.shape[0]), voxels=int(cen.shape[0]), vox_res=320, bytes=raw.shape[0] * (12 * 4 + 8) + 320 ** 3 * 16)
vox["GBps"] = vox["bytes"] / dt / 1e9
# grid (brick map) build from scratch -- what every prune / grow event costs (the reference rebuilds its hash on EVERY query)
from pointnerf_amd.point_query import lighting_fast_querier, clear_grid_cache
grid_ms = {}
for name, (label, make_opt, points_fn, n_pts, rays_fn) in bench._cfg().items():
    if name == "chair":
        continue
    o2 = make_opt(is_train=0)
    x2 = torch.from_numpy(points_fn(n_pts)).to(dev)
    q2 = lighting_fast_querier(dev, o2)
    d2 = rays_fn(0, 4096)
    rd2, cp2 = torch.from_numpy(d2["raydir"]).to(dev), torch.from_numpy(d2["campos"]).to(dev)
    for rep in range(3):
        clear_grid_cache()
        ops.prof_collect()
        q2.query_dense(x2[None], x2.shape[0], float(d2["near"].min()), float(d2["far"].max()), rd2, cp2)
        torch.cuda.synchronize()
        pr = ops.prof_collect()
    gi = q2.last_grid_info
    grid_ms[name] = dict(points=int(n_pts), build_ms=pr["grid_build"][0] / max(pr["grid_build"][1], 1), occupied_cells=int(gi["n_occ"]), max_points_per_cell=int(gi["max_cnt"]))
    del x2, q2
for k, v in res.items():
    v["GBps"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
    v["frac_of_8TBps"] = v["GBps"] / 8000.0
res["full_image_800x800"] = full
res["voxel_downsample_2M_res320"] = vox
res["grid_build"] = grid_ms
print(json.dumps(res, indent=1))
