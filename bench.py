"""bench.py -- rays/sec (render + backward) of the Point-NeRF hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (default, and what the driver runs) = BASELINE.json configs[1]: synthetic NeRF-synthetic 'lego' (2M neural points,
800x800 poses, K=8, 128 samples/ray, lego_cuda.sh values for everything else).  `--config chair | scannet | barn`
benches the other single-GPU configurations of BASELINE.json (configs[0], [3], [4]: same step, their own scene generator and
script values); configs[2] is `--gpus 8` of the default.  One "step" is one optimisation step of
the hot path over one batch of `--rays` rays per GPU, inputs already resident in HBM:
    query (grid cached: xyz is fixed) -> aggregator MLP -> ray-march -> masked MSE + conf regulariser
    -> backward -> [N>1: RCCL all-reduce of the gradients] -> 2x Adam (MLP lr, points plr) .
N>1 shards the rays of ONE global batch of `--rays` rays contiguously across the ranks (BASELINE.json configs[2] as SURVEY 8d defines it:
"scaling": "strong"), point cloud and MLP replicated; `--weak` keeps `--rays` rays PER GPU instead (the optional weak-scaling variant);
the only collectives are the gradient all-reduce and a 2-float loss normaliser.
Rank 0 prints ONE JSON line; `value` is whole-job rays/sec.  `roofline` (dominant kernel): `frac` = SURVEY 8d algorithmic work / nominal peak,
`frac_executed` = executed f16 products / nominal peak, `peak_measured` = the ceiling timed in this run (f16 MFMA with toggling operands:
pnerf_debug_mfma_rate; HBM entries: a 1 GiB device copy), `traffic` = PMC bytes of the committed profile pass (stamped with its commit).
`cpu_baseline`: kind "port" (oracle) on the GPU box, "reference" where /root/reference imports (`--cpu-baseline-only`).
`--force-collectives`: one-rank RCCL bring-up of every collective of the step; N > 1 runs are guarded by watchdogs that leave a record.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_ROW_FWD = 542720            # per valid neighbor row, forward (2*(284*256+256*256+263*256+256*256+256))
FLOP_SAMPLE_FWD = 137984         # per valid sample, colour MLP forward
FLOP_ROW_DGRAD = 2 * (256 * 256 + 256 * 263 + 256 * 256 + 224 * 256) + 2 * 256   # dY @ W for the four layers (d X0: the 224 embedding columns) + alpha head
FLOP_ROW_WGRAD = 2 * (284 * 256 + 256 * 256 + 263 * 256 + 256 * 256)             # dY^T X GEMMs
FLOP_SAMPLE_WGRAD = 2 * (280 * 128 + 2 * 128 * 128)
# the four 256-wide aggregator layers run on v_mfma_f32_32x32x16_f16 with two-plane operands: THREE f16 products per algorithmic
# multiply-add (csrc/f16x3.h), so the matrix pipe executes 3x the algorithmic flops; its roofline is the dense f16 peak
F16_PRODUCTS = 3
# csrc/mixq.h (round 6): the leading product on f16 factors + the two cross terms on e4m3 factors at half the pipe cycles each = 2 f16-product
# equivalents on the 256 columns of a layer that run the mixed format (its tail columns, 16 / 32 of layers 3 / 1 of the forward, keep three)
MIX_PRODUCTS_BWD = 2.0
MIX_PRODUCTS_FWD = (4 * 256 * 2.0 + 48 * 3.0) / (4 * 256 + 48)
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense BF16/F16 MFMA (measured 2178-2495)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense (no kernel of the path uses it any more)
PEAK_HBM_GBS = 8000.0
# roofline.traffic is NOT measured inside a bench run: it is read from the committed PMC summary (separate rocprofv3 --pmc passes of this
# same command, FETCH x 2 + WRITE per the guide; tools/gpu_round_profile.sh regenerates it)
TRAFFIC_SOURCE = "profiles/traffic.json (static: rocprofv3 --pmc passes of this command, not measured in this run)"


def measured_copy_gbs(dev, nbytes=1 << 30, reps=6):
    """SURVEY 8d's HBM denominator beside the nominal 8 TB/s: a device-to-device copy of `nbytes` (16-byte accesses, read + write counted) timed with
    events on the current stream, best of `reps` after one untimed pass"""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a); torch.mul(a, 1.0, out=b)
    best = float("inf")
    for i in range(2 * reps):              # the runtime's copy and a vectorised elementwise kernel (x * 1.0): whichever is faster is the ceiling
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if i % 2:
            b.copy_(a)
        else:
            torch.mul(a, 1.0, out=b)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del a, b
    return 2.0 * nbytes / (best * 1e-3) / 1e9
# algorithmic HBM bytes of the weight-gradient GEMMs per neighbor row, read once: the inputs of layers 2 .. 4 as ONE f16 plane
# (2 B x (256 + 288 + 256)), layer 1's input as its saved last 64 columns (2 B x 64) + the 128-byte embedding row + 16 B of row
# metadata it is rebuilt from (k_wgrad_x0), and the four output gradients as one f16 plane (2 B x 4 x 256)
BYTES_ROW_WGRAD = 2 * (256 + 288 + 256) + (2 * 64 + 128 + 16) + 2 * 4 * 256
# ... of the colour MLP's three weight-gradient GEMMs per valid sample ([f | view encoding] 288, c1, c2 and d c1..d c3, one plane each)
BYTES_SAMPLE_WGRAD = 2 * (288 + 128 + 128) + 2 * 3 * 128
# ... of the training forward (gather 168 B + the saved planes: X0's last 64 columns, h1 256, [h2|extras] 288, h3 256 as one plane, h4 256
# columns as two + row metadata)
BYTES_ROW_FWD = 168 + 2 * (64 + 256 + 288 + 256) + 2 * 2 * 256 + 16 + 4 + 96
# ... of the backward (h4 planes + sign words + metadata + d f and embedding rows read; four dY planes written)
BYTES_ROW_BWD = 2 * 2 * 256 + 96 + 20 + 128 + 128 + 2 * 4 * 256

def _cfg():
    from pointnerf_amd import config, scenes
    return {
        # name: (BASELINE.json entry, opt, point generator, default point count, ray generator(step, R))
        "chair": ("configs[0]: synthetic chair, 64x64 crop of an 800x800 view", config.chair_opt, scenes.chair_points, 8192,
                  lambda i, R: scenes.block_rays(theta_deg=30.0 + 3.6 * i, size=max(1, int(R ** 0.5)))),
        "lego": ("configs[1]: synthetic lego, 800x800 poses", config.bench_lego_opt, scenes.lego_points, 2_000_000, scenes.random_rays),
        "scannet": ("configs[3]: ScanNet-scale room, 640x480 poses", config.scannet_opt, scenes.scannet_points, 6_000_000, scenes.scannet_rays),
        "barn": ("configs[4]: Barn-scale shell, 1088x640 poses", config.barn_opt, scenes.barn_points, 20_000_000, scenes.barn_rays),
    }


CONFIGS = ("barn", "chair", "lego", "scannet")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="lego", choices=sorted(CONFIGS), help="BASELINE.json configuration (default: configs[1], the headline)")
    ap.add_argument("--rays", type=int, default=65536, help="rays of the global batch per step (N > 1: split contiguously over the ranks; with --weak: rays per GPU)")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling (--rays rays per GPU, global batch N x --rays) instead of the default strong scaling (configs[2])")
    ap.add_argument("--cross-terms", type=int, default=8, choices=(8, 16), help="cross terms of the aggregator's tile GEMMs: 8 = e4m3 (csrc/mixq.h, default), 16 = f16 (csrc/f16x3.h)")
    ap.add_argument("--cross-terms-where", type=int, default=None, help="bit mask of the kernels that use the e4m3 cross terms: 1 inference forward, 2 training forward, 4 backward (library default: 4)")
    ap.add_argument("--no-variants", action="store_true", help="skip the supplementary variants of the default run (f16 cross terms, e4m3 forward, reference shell, fp32-class weight gradients)")
    ap.add_argument("--points", type=int, default=0, help="neural points (0 = the configuration's own count)")
    ap.add_argument("--cpu-rays", type=int, default=12288, help="rays of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--render-only", action="store_true", help="supplementary: inference (render, no backward) rays/s")
    ap.add_argument("--inference-products", type=int, default=3, choices=(2, 3), help="--render-only: MFMA products per multiply-add of the inference forward (3 = fp32-class, default; 2 = weights' high plane only)")
    ap.add_argument("--no-overlap-comm", action="store_true", help="N > 1: all-reduce the point gradients after the whole backward instead of behind the weight-gradient GEMMs")
    ap.add_argument("--point-grads", default="auto", choices=("auto", "dense", "sparse"),
                    help="N > 1: exchange of the per-point gradients: dense all-reduce (one bucket, overlapped), sparse touched-row exchange, auto = sparse from 6 M points")
    ap.add_argument("--unfused-color-loss", action="store_true", help="A/B: the colour loss as ATen ops on the compacted hit rays (argsort + index_selects)")
    ap.add_argument("--unfused-zero-one", action="store_true", help="A/B: the zero-one regulariser as the reference's chain of ATen ops on a materialised conf_coefficient")
    ap.add_argument("--zero1", action="store_true", help="N > 1: shard the point-parameter Adam (reduce-scatter + all-gather) instead of all-reducing the gradients")
    ap.add_argument("--wgrad-planes", type=int, default=1, choices=(1, 2),
                    help="f16 planes per operand of the weight-gradient GEMMs: 1 = shipped (one plane rounded to nearest, one product); 2 = both operands "
                         "as two planes, three products (fp32-class weight gradients).  The default run times 1 and ALSO reports 2 as config.fp32_class_variant")
    ap.add_argument("--no-fp32-class-variant", action="store_true", help="skip the supplementary --wgrad-planes 2 measurement of the default run")
    ap.add_argument("--force-collectives", action="store_true",
                    help="bring-up on a 1-GPU box: --gpus 1 with a one-rank RCCL group and EVERY collective of the step executed (as identities): the "
                         "calls, streams and in-place forms the 8-GPU run will make, on the real backend")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="no GPU: time only the cpu_baseline leg (kind reference where /root/reference imports, else port) and print it")
    return ap.parse_args()


def build_model(opt, n_points, dev, points_fn=None):
    from pointnerf_amd import scenes
    from pointnerf_amd.neural_points import NeuralPoints
    from pointnerf_amd.point_aggregators import PointAggregator
    from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
    xyz = torch.from_numpy((points_fn or scenes.lego_points)(n_points)).to(dev)
    attrs = {k: torch.from_numpy(v).to(dev) for k, v in scenes.point_attributes(xyz.shape[0], opt.point_features_dim, 1).items()}
    torch.manual_seed(0)                                   # identical random-init weights on every rank (replicated)
    agg = PointAggregator(opt).to(dev)
    agg.flatten_()
    npnt = NeuralPoints(opt.point_features_dim, xyz.shape[0], opt, dev)
    npnt.set_points(xyz, attrs["points_embeding"], points_color=attrs["points_color"], points_dir=attrs["points_dir"],
                    points_conf=attrs["points_conf"], parameter=True)
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt).to(dev)
    agg.flatten_()
    return model


def step_inputs(step, rank, world, rays, dev, rays_fn=None):
    """Rank `rank`'s contiguous slice of the global batch of world*rays random pixels of train-like pose `step`."""
    from pointnerf_amd import scenes
    d = (rays_fn or scenes.random_rays)(step % 100, rays * world)
    sl = slice(rank * rays, (rank + 1) * rays)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(campos=t(d["campos"]), camrotc2w=t(d["camrotc2w"]), raydir=t(d["raydir"][:, sl]), gt_image=t(d["gt_image"][:, sl]),
                near=t(d["near"]), far=t(d["far"]), bg_color=t(d["bg_color"]), pixel_idx=t(d["pixel_idx"][:, sl]))


def loss_fn(opt, out, inp, world):
    from pointnerf_amd import dist as pdist
    return pdist.hot_path_loss(opt, out, inp["gt_image"])


def cpu_baseline(opt, n_points, rays, threads, points_fn=None, rays_fn=None):
    """The oracle (C restatement of the query + torch-CPU restatement of aggregator/ray-march, `kind: port`) on a
    bounded sample of the SAME workload: the first `rays` rays of step 0, forward + loss + backward."""
    from pointnerf_amd import scenes
    from oracle import pyref
    torch.set_num_threads(threads)
    xyz = torch.from_numpy((points_fn or scenes.lego_points)(n_points))
    attrs = {k: torch.from_numpy(v).requires_grad_(True) for k, v in scenes.point_attributes(xyz.shape[0], 32, 1).items()}
    mlp = {k: v.requires_grad_(True) for k, v in pyref.init_mlp_params(opt, seed=0).items()}
    d = (rays_fn or scenes.random_rays)(0, 65536)
    d["raydir"], d["gt_image"] = d["raydir"][:, :rays], d["gt_image"][:, :rays]
    inp = pyref.to_torch_inputs(d)
    # the serial voxel-grid build over all points (the C oracle rebuilds it on every query, like the reference): timed on its own with a
    # one-ray query, so that the sample's seconds can be read with and without it
    one = dict(inp)
    one["raydir"], one["gt_image"] = inp["raydir"][:, :1], inp["gt_image"][:, :1]
    t0 = time.time()
    pyref.query(opt, xyz, one, nthreads=threads)
    t_grid = time.time() - t0
    t0 = time.time()
    q = pyref.query(opt, xyz, inp, nthreads=threads)
    t_query = time.time() - t0
    t0 = time.time()
    out = pyref.render(opt, dict(xyz=xyz, **attrs), mlp, inp, q=q, nthreads=threads)
    loss = pyref.training_loss(opt, out, inp)
    loss.backward()
    t_render = time.time() - t0
    dt = t_query + t_render
    return dict(value=rays / dt, unit="rays/s", cores=threads, kind="port",
                value_without_grid_build=rays / max(dt - t_grid, 1e-9),
                seconds={"grid_build_serial": t_grid, "query_incl_grid_build": t_query, "aggregator_raymarch_loss_backward": t_render},
                sample="first %d rays of step 0 of the same workload: query %.1f s (of which the serial grid build over %d points %.1f s), "
                       "aggregator + ray-march forward, loss, backward %.1f s" % (rays, t_query, xyz.shape[0], t_grid, t_render))


def cpu_baseline_reference(opt, n_points, rays, threads, points_fn=None, rays_fn=None):
    """`kind: reference`: the REFERENCE'S OWN modules on the host cores, where /root/reference exists (the authoring container; it does not
    travel to the GPU box) -- the native query op as the reference's kernels compiled for the host (oracle/_ref/libref_query.so: query_worldcoords.cu run
    block by block, serial, K <= 8), models/aggregators/point_aggregators.py PointAggregator.forward (:727-814), models/rendering/diff_ray_marching.py
    ray_march, the training loss and torch autograd's backward through them.  Same bounded sample as the port."""
    import argparse as _ap
    from pointnerf_amd import scenes
    from oracle import pyref
    # (a `models` package may already be in sys.modules -- pointnerf_amd's overlay installs one: drop it, import the reference's, and check where
    #  every module came from before the result is labelled 'reference')
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    if sys.path[0] != "/root/reference":
        sys.path.insert(0, "/root/reference")
    import models.aggregators.point_aggregators as _ref_pa                                  # (reference)
    import models.rendering.diff_ray_marching as _ref_rm                                    # (reference)
    import models.rendering.diff_render_func as _ref_rf                                     # (reference)
    for _m in (_ref_pa, _ref_rm, _ref_rf):
        if not os.path.abspath(_m.__file__).startswith("/root/reference/"):
            raise RuntimeError("%s was imported from %s, not from the reference checkout" % (_m.__name__, _m.__file__))
    PointAggregator, ray_march = _ref_pa.PointAggregator, _ref_rm.ray_march
    find_render_function, find_blend_function = _ref_rf.find_render_function, _ref_rf.find_blend_function
    torch.set_num_threads(threads)
    p = _ap.ArgumentParser()
    PointAggregator.modify_commandline_options(p, True)
    ro = p.parse_args([])
    for k, v in vars(opt).items():
        setattr(ro, k, v)
    ro.agg_axis_weight = None
    xyz = torch.from_numpy((points_fn or scenes.lego_points)(n_points))
    attrs = {k: torch.from_numpy(v).requires_grad_(True) for k, v in scenes.point_attributes(xyz.shape[0], 32, 1).items()}
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # (the reference's constructor prints; stdout carries the one JSON line)
        agg = PointAggregator(ro)
    agg.load_state_dict(pyref.init_mlp_params(opt, seed=0), strict=True)
    d = (rays_fn or scenes.random_rays)(0, 65536)
    d["raydir"], d["gt_image"] = d["raydir"][:, :rays], d["gt_image"][:, :rays]
    inp = pyref.to_torch_inputs(d)
    t0 = time.time()
    q = pyref.query(opt, xyz, inp, impl="ref")
    t_query = time.time() - t0
    t0 = time.time()
    points = dict(xyz=xyz, **attrs)
    nb = pyref.gather_neighbors(points, q["sample_pidx"], inp["camrotc2w"][0], inp["campos"][0])        # (the gather of neural_points.py:690-717)
    out, ray_valid, weight, conf_c = agg(nb["color"], torch.eye(3), nb["dir"], nb["conf"], nb["emb"], nb["xyz_pers"], nb["xyz"], nb["mask"],
                                         q["sample_loc"], q["sample_loc_w"], q["sample_ray_dirs"], q["hp"]["vsize"], 0)
    rd = pyref.ray_dist(opt, q["sample_loc"], ray_valid)
    color, _, opacity, acc, bw, bg_t, _ = ray_march(rd, ray_valid, out, find_render_function("radiance"), find_blend_function("alpha"), inp["bg_color"])
    loss = pyref.training_loss(opt, dict(ray_mask=q["ray_mask"], coarse_raycolor=color, conf_coefficient=conf_c), inp)
    loss.backward()
    t_render = time.time() - t0
    dt = t_query + t_render
    return dict(value=rays / dt, unit="rays/s", cores=threads, kind="reference",
                seconds={"query_reference_kernels_serial_incl_grid_build": t_query, "aggregator_raymarch_loss_backward": t_render},
                sample="first %d rays of step 0 of the same workload through the reference's modules: native query (reference kernels on the host, 1 core) "
                       "%.1f s, PointAggregator.forward + ray_march + loss + backward (torch CPU, %d threads) %.1f s" % (rays, t_query, threads, t_render))


class Watchdog:
    """A first contact with RCCL that hangs must leave a record, not a driver timeout: if the guarded block has not finished after `seconds`,
    every rank writes what it knows (stage, rank, backend, library version, the NCCL_ / RCCL_ / HSA_ environment, the hint to rerun with
    NCCL_DEBUG=INFO) to stderr and the process exits with code 17."""

    def __init__(self, what, seconds, rank, world):
        self.what, self.seconds, self.rank, self.world = what, seconds, rank, world

    def _fire(self):
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:       # noqa: BLE001
            ver = "unknown (%r)" % (e,)
        env = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "MASTER_", "HIP_VISIBLE", "ROCR_VISIBLE")) or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        print(json.dumps({"bench_watchdog": "%s did not finish within %d s" % (self.what, self.seconds), "rank": self.rank, "world": self.world,
                          "rccl_version": ver, "env": env,
                          "hint": "rerun with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL (RCCL reads the NCCL_ variables); HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this pool"}),
              file=sys.stderr, flush=True)
        os._exit(17)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def rccl_selftest(dev, rank, world):
    """One small instance of every collective form the step uses, before anything is timed: a 1 KB all-reduce, an all_gather_into_tensor
    (gloo: all_gather), a reduce_scatter_tensor, and the side-stream in-place all-reduce of a bucket head behind an event with
    record_stream (dist.allreduce_grads).  Raises on a wrong sum; rank 0 reports on stderr (stdout carries the one JSON line)."""
    import torch.distributed as dist
    backend = dist.get_backend()
    x = torch.full((256,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    want = world * (world + 1) / 2.0
    assert float(x[0]) == want and float(x[-1]) == want, "all_reduce: %r != %r" % (float(x[0]), want)
    mine = torch.full((64,), float(rank), device=dev)
    allv = torch.empty(64 * world, device=dev)
    if backend == "gloo":
        dist.all_gather(list(allv.view(world, 64).unbind(0)), mine)
    else:
        dist.all_gather_into_tensor(allv, mine)
    assert allv.view(world, 64)[:, 0].tolist() == [float(r) for r in range(world)], "all_gather_into_tensor"
    if backend != "gloo":
        buf = torch.arange(world * 32, device=dev, dtype=torch.float32)
        out = buf[rank * 32:(rank + 1) * 32]
        dist.reduce_scatter_tensor(out, buf)                       # in place, as optim.ShardedAdam does
        assert float(out[0]) == world * rank * 32.0, "reduce_scatter_tensor"
    bucket = torch.full((1 << 16,), float(rank + 1), device=dev)
    ev = torch.cuda.Event()
    ev.record()
    comm = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(comm):
        comm.wait_event(ev)
        bucket.record_stream(comm)
        dist.all_reduce(bucket[:1 << 15])
    torch.cuda.current_stream(dev).wait_stream(comm)
    assert float(bucket[0]) == want and float(bucket[-1]) == float(rank + 1), "side-stream bucket all-reduce"
    torch.cuda.synchronize()
    if rank == 0:
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:            # noqa: BLE001
            ver = "unknown"
        print("rccl_selftest: ok  backend=%s world=%d rccl=%s HSA_ENABLE_IPC_MODE_LEGACY=%s NCCL_DEBUG=%s (set NCCL_DEBUG=INFO for RCCL's own log)"
              % (backend, world, ver, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), os.environ.get("NCCL_DEBUG")), file=sys.stderr, flush=True)
    return "ok backend=%s world=%d" % (backend, world)


def best_cpu_baseline(opt, n_points, rays, threads, points_fn=None, rays_fn=None):
    """`reference` when the reference's modules import (only where /root/reference exists), the port otherwise (the GPU box)"""
    if os.path.isdir("/root/reference/models/aggregators"):
        try:
            from oracle import query as oq
            oq.build()
            return cpu_baseline_reference(opt, n_points, rays, threads, points_fn, rays_fn)
        except Exception as e:       # noqa: BLE001
            print("cpu_baseline: the reference's modules did not run (%r): timing the port" % (e,), file=sys.stderr)
    return cpu_baseline(opt, n_points, rays, threads, points_fn, rays_fn)


def main():
    args = parse()
    if args.cpu_baseline_only:
        cfg_name, opt_fn, points_fn, n_default, rays_fn = _cfg()[args.config]
        print(json.dumps(best_cpu_baseline(opt_fn(is_train=1), args.points or n_default, args.cpu_rays, min(os.cpu_count() or 1, 32), points_fn, rays_fn)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # --gpus N without N ranks would print an N = 1 number under an N-GPU label, with no collective ever exercised
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d"
                         % (args.gpus, world, args.gpus, args.gpus))
    dist_on = world > 1 or args.force_collectives
    if args.force_collectives:
        if "MASTER_PORT" not in os.environ:             # a port nobody holds (a fixed one can collide with a neighbouring run)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        # "nccl" IS RCCL on ROCm.  PNERF_DIST_BACKEND=gloo exists only so that tests can run 2 ranks on a 1-GPU box.
        with Watchdog("init_process_group", int(os.environ.get("PNERF_INIT_TIMEOUT", "300")), rank, world):      # (ranks of a fresh box may reach this a minute apart: first import of torch)
            torch.distributed.init_process_group(backend=os.environ.get("PNERF_DIST_BACKEND", "nccl"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation (oracle/ is the checker only)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from pointnerf_amd import ops, dist as pdist
    from pointnerf_amd.fused import FusedRender
    selftest = None
    pdist.FORCE_COLLECTIVES = bool(args.force_collectives)
    if dist_on:
        with Watchdog("the collective self-test (first RCCL communicator + 4 small collectives)", int(os.environ.get("PNERF_SELFTEST_TIMEOUT", "180")), rank, world):
            selftest = rccl_selftest(dev, rank, world)       # raises on a wrong sum
        if not selftest:
            raise SystemExit("bench.py: the collective self-test did not run: refusing to time a multi-GPU step")
    ops.set_wgrad_planes(args.wgrad_planes)
    ops.set_cross_terms(args.cross_terms, where=args.cross_terms_where)
    # N > 1: a rank that stops inside a later collective (a mismatch between the ranks' call sequences) also leaves a record
    # (re-armed per phase: a slow but healthy run -- a cold first build, the Barn cloud on 8 ranks -- is not killed by one global budget)
    guard_s = int(os.environ.get("PNERF_RUN_TIMEOUT", "1500"))
    run_guard = Watchdog("model build + set-up", guard_s, rank, world) if dist_on else None
    if run_guard is not None:
        run_guard.__enter__()

    def rearm(what):
        nonlocal run_guard
        if run_guard is not None:
            run_guard.__exit__(None, None, None)
            run_guard = Watchdog(what, guard_s, rank, world)
            run_guard.__enter__()
    try:
        _run(args, world, rank, dev, dist_on, selftest, rearm)
    finally:
        if run_guard is not None:
            run_guard.__exit__(None, None, None)
    if dist_on:
        torch.distributed.destroy_process_group()


def _run(args, world, rank, dev, dist_on, selftest, rearm):
    from pointnerf_amd import ops, dist as pdist
    from pointnerf_amd.fused import FusedRender

    # is_train=1: the reference trains with 30 % segment jitter (point_query.py:81); the in-kernel RNG path is what a
    # training step runs, so it is what is timed (parity runs -- tests/ -- use jitter off, where results are bit-defined)
    cfg_name, opt_fn, points_fn, n_default, rays_fn = _cfg()[args.config]
    n_points = args.points or n_default
    if args.config == "chair":
        args.rays = min(args.rays, 4096)               # configs[0] is a 64x64 crop
    strong = world > 1 and not args.weak                # configs[2]: the identical global batch split over the ranks
    rays_rank = args.rays // world if strong else args.rays
    if strong and rays_rank * world != args.rays:
        raise SystemExit("bench.py: --rays %d is not a multiple of --gpus %d" % (args.rays, world))
    opt = opt_fn(is_train=0 if args.render_only else 1)
    if args.render_only:
        ops.set_inference_products(args.inference_products)
    model = build_model(opt, n_points, dev, points_fn)
    agg, npnt = model.aggregator, model.neural_points
    model.fused_zero_one = not args.unfused_zero_one     # the zero-one regulariser as one fused pass over the neighbor table (ops.ZeroOneConf)
    model.fused_color_loss = not args.unfused_color_loss and not args.unfused_zero_one   # the colour loss over the dense ray colours (ops.ColorLossRays)
    mlp_params = [p for p in agg.parameters() if p.requires_grad]
    pt_params = [p for p in (npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color) if p.requires_grad]
    # the reference's two Adam instances (mvs_points_volumetric_model.py:80-91) as one-pass HIP updates; --zero1 shards the
    # update and its state over the ranks (reduce-scatter / all-gather instead of all-reduce)
    from pointnerf_amd.optim import FusedAdam, ShardedAdam, step_all
    zero1 = args.zero1 and dist_on
    sparse = dist_on and not zero1 and (args.point_grads == "sparse" or (args.point_grads == "auto" and n_points >= 6_000_000))
    opt_mlp = FusedAdam(mlp_params, lr=opt.lr, betas=(0.9, 0.999))
    opt_pts = ShardedAdam(pt_params, lr=opt.plr, betas=(0.9, 0.999)) if zero1 else FusedAdam(pt_params, lr=opt.plr, betas=(0.9, 0.999))

    comm_marks = None                         # (event pairs around the gradient exchange on the main stream; filled in the timed region)
    total = args.warmup + args.steps
    inputs = [step_inputs(i, rank, world, rays_rank, dev, rays_fn) for i in range(total)]   # resident in HBM before timing

    def one_step(inp):
        nonlocal comm_marks
        if args.render_only:
            with torch.no_grad():
                out = model(**inp)
            return out["coarse_raycolor"].sum(), model.last_stats
        opt_mlp.zero_grad(set_to_none=True); opt_pts.zero_grad(set_to_none=True)
        out = model(**inp)
        loss = loss_fn(opt, out, inp, world)
        loss.backward()
        if dist_on and comm_marks is not None:
            comm_marks.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            comm_marks[-1][0].record()
        # no-op at N=1; RCCL over xGMI otherwise.  The three point tensors only the renderer writes (88 % of the bytes) start
        # their all-reduce as soon as the input-gradient kernels are done, under the weight-gradient GEMMs
        if sparse:
            # large clouds: a rank's rays touch a few percent of the points -- exchange the touched rows only (dist.sparse_allreduce_rows);
            # the row list and the largest count over the ranks were prepared right after the query (model.after_query) and read with the
            # step's one host read, so nothing in here synchronises
            ids, cnt, cap = model.sparse_plan
            pdist.sparse_allreduce_rows([p.grad for p in pt_params], ids[:cnt].long(), cap=cap)
            pdist.allreduce_grads(mlp_params, [])
        else:
            early = [] if (zero1 or args.no_overlap_comm) else [npnt.points_embeding, npnt.points_dir, npnt.points_color]
            pdist.allreduce_grads(mlp_params, [] if zero1 else pt_params, ready_event=FusedRender.point_grads_ready if early else None, early_params=early)
        if dist_on and comm_marks is not None:
            comm_marks[-1][1].record()        # main stream: backward done -> every gradient summed = the communication that did NOT hide
        step_all([opt_mlp, opt_pts])          # both Adam instances in one launch (ShardedAdam, when --zero1, steps on its own)
        return loss, model.last_stats

    # the 15 camera numbers per batch the host passes to the library by value (position, rotation, background, near / far): their host
    # copies are made here, once per input tensor (ops.host_array caches them on the tensor object), not as synchronisations inside steps
    for inp in inputs:
        for k in ("campos", "camrotc2w", "bg_color", "near", "far"):
            ops.host_array(inp[k])
    if not args.render_only:
        # size the activation arena for the largest of the batches that will be run (their neighbor tables differ: up to 2x between
        # poses of the Barn-scale configuration); the query alone tells the number of valid samples
        from pointnerf_amd import _lib as L
        biggest = 0
        with torch.no_grad():
            for inp in inputs:
                dense = npnt.query_dense(inp)              # (the jitter draws differ from the timed steps': the counts agree to a percent, the reserve has 15 % headroom)
                biggest = max(biggest, int(dense["counters"][0].item()))
        need = int(L.lib().pnerf_agg_saved_bytes(biggest, int(opt.K)))
        if need <= ops.arena_budget_bytes():
            ops.ARENA.reserve(int(need * 1.05), dev)
    if sparse:
        model.plan_sparse = lambda dense: pdist.plan_sparse_exchange(dense["sample_pidx"], n_points)
    stats = []
    # set-up, not warm-up: two untimed steps on the first batch size the activation arena (one ~80 GB hipMalloc) and let the caching
    # allocator settle, so that the measurement does not depend on how many warm-up steps the caller asks for
    for _ in range(2):
        one_step(inputs[0])
    for i in range(args.warmup):
        one_step(inputs[i])
    torch.cuda.synchronize()
    rearm("the timed steps")
    if not args.no_prof:
        ops.prof_enable(True)
        ops.prof_collect()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # step boundaries on the stream the kernels run on
    comm_marks = [] if dist_on else None
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.warmup, total):
        loss, st = one_step(inputs[i])
        marks[i - args.warmup + 1].record()
        stats.append(st)
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    extra_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0      # hipMalloc calls inside the timed region
    prof = None
    if not args.no_prof:
        prof = ops.prof_collect()
        ops.prof_enable(False)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    dt_local = dt
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist_on:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    # supplementary, outside the timed region: one step with the voxel grid rebuilt (what the reference pays every
    # step: query_worldcoords.cu:308-365), and one step without the optimizer
    extra = {}
    if not args.render_only:
        from pointnerf_amd import point_query
        def timed(fn):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            return (time.perf_counter() - t) * 1e3
        def cold():
            point_query.clear_grid_cache()
            one_step(inputs[-1])
        def no_adam():
            opt_mlp.zero_grad(set_to_none=True); opt_pts.zero_grad(set_to_none=True)
            loss_fn(opt, model(**inputs[-1]), inputs[-1], world).backward()
        extra = {"ms_step_cold_grid": timed(cold), "ms_step_without_optimizer": timed(no_adam)}
        # ---- variants of the SAME step, outside the timed region (each: two settling steps, then up to 10 timed ones)
        def variant(what, setup, restore, step_fn=None):
            step_fn = step_fn or one_step
            setup()
            try:
                for _ in range(2):
                    step_fn(inputs[0])
                nv = min(args.steps, 10)
                torch.cuda.synchronize(); tv0 = time.perf_counter()
                for i in range(nv):
                    lv, _ = step_fn(inputs[args.warmup + i])
                torch.cuda.synchronize()
                msv = (time.perf_counter() - tv0) / nv * 1e3
            finally:
                restore()
            return {"ms_per_step": msv, "value": rays_rank / (msv * 1e-3), "unit": "rays/s", "steps": nv, "final_loss": float(lv.item()), "what": what}

        if not dist_on and not args.no_variants and args.cross_terms == 8 and args.cross_terms_where is None:
            # (a) f16 cross terms everywhere: the arithmetic of rounds 2-5 (csrc/f16x3.h: three f16 products per multiply-add in forward and backward)
            extra["f16_cross_terms_variant"] = variant(
                "pnerf_set_cross_terms(16): f16 cross terms in every tile GEMM (three f16 products per multiply-add, the round-5 arithmetic); the headline runs "
                "the backward's input-gradient chain with e4m3 cross terms (csrc/mixq.h)",
                lambda: ops.set_cross_terms(16), lambda: ops.set_cross_terms(8))
            # (b) e4m3 cross terms in the training forward as well: sigma / RGB 1e-5 .. 6e-5 from the oracle (bar 1e-4, tests/test_gpu_mix.py), NOT the
            # default because LeakyReLU pre-activations that close to zero take the other branch than the fp32 oracle's (gradient comparisons see it)
            extra["e4m3_forward_variant"] = variant(
                "pnerf_set_cross_terms_where(7): e4m3 cross terms in the training forward too (sigma / RGB within 6e-5 of the oracle instead of 1e-6: inside "
                "north_star's 1e-4, but the LeakyReLU masks then differ from the fp32 oracle's near zero -- not the default)",
                lambda: ops.set_cross_terms(8, where=7), lambda: ops.set_cross_terms(8, where=4))
            # (c) the route the reference's UNMODIFIED shell takes through the overlay (models/mvs_points_volumetric_model.py:98-118, base_rendering_model.py
            # :533-662): the losses as ATen chains on the compacted outputs and two torch.optim.Adam instances
            shell = {}

            def shell_setup():
                shell["flags"] = (model.fused_zero_one, model.fused_color_loss)
                model.fused_zero_one = model.fused_color_loss = False
                shell["mlp"] = torch.optim.Adam(mlp_params, lr=opt.lr, betas=(0.9, 0.999))
                shell["pts"] = torch.optim.Adam(pt_params, lr=opt.plr, betas=(0.9, 0.999))

            def shell_restore():
                model.fused_zero_one, model.fused_color_loss = shell["flags"]

            def shell_step(inp):
                shell["mlp"].zero_grad(set_to_none=True); shell["pts"].zero_grad(set_to_none=True)
                out = model(**inp)
                loss = loss_fn(opt, out, inp, world)
                loss.backward()
                shell["mlp"].step(); shell["pts"].step()
                return loss, model.last_stats
            extra["reference_shell_variant"] = variant(
                "the step as the reference's unmodified mvs_points_volumetric_model.py runs it through the overlay (INTEGRATION.md 2): compacted outputs, the colour "
                "loss and the zero-one regulariser as ATen chains (--unfused-color-loss --unfused-zero-one), two torch.optim.Adam instances",
                shell_setup, shell_restore, shell_step)
        # the SAME step with fp32-class weight gradients (both operands of every weight-gradient GEMM as two f16 planes, three products:
        # pnerf_set_wgrad_planes(2)) -- the headline's arithmetic caveat priced in the same run, outside the timed region
        if args.wgrad_planes == 1 and not dist_on and not args.no_fp32_class_variant and not args.no_variants:
            ops.set_wgrad_planes(2)
            try:
                need2 = int(L.lib().pnerf_agg_saved_bytes(biggest, int(opt.K)))
                if need2 <= ops.arena_budget_bytes():
                    ops.ARENA.reserve(int(need2 * 1.05), dev)
                for _ in range(2):
                    one_step(inputs[0])
                nv = min(args.steps, 10)
                torch.cuda.synchronize(); tv0 = time.perf_counter()
                for i in range(nv):
                    lv, _ = one_step(inputs[args.warmup + i])
                torch.cuda.synchronize()
                msv = (time.perf_counter() - tv0) / nv * 1e3
                extra["fp32_class_variant"] = {"ms_per_step": msv, "value": rays_rank / (msv * 1e-3), "unit": "rays/s", "steps": nv,
                                               "final_loss": float(lv.item()),
                                               "what": "the same step with --wgrad-planes 2: weight-gradient GEMM operands as 2 x f16 planes (22 bits), three "
                                                       "products per multiply-add -- the arithmetic of the forward and of the input-gradient chain; "
                                                       "tests/test_gpu_zz_convergence.py, tests/test_gpu_bench_config.py compare the two"}
            finally:
                ops.set_wgrad_planes(1)
    if not np.isfinite(float(loss.item())) and not os.environ.get("PNERF_BENCH_ALLOW_NAN"):      # (the env switch exists for dev variants that drop work on purpose)
        raise SystemExit("bench.py: non-finite loss after %d steps -- the timed path produced NaN/Inf, the number would be meaningless" % total)
    per_rank_ms = [dt / args.steps * 1e3]
    exposed_ms = replica_spread = None
    if dist_on:                                     # self-check for the scaling record: the RCCL world and every rank's own step time
        t_all = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        torch.distributed.all_gather(t_all, torch.tensor([dt_local / args.steps * 1e3], device=dev, dtype=torch.float64))
        per_rank_ms = [float(t.item()) for t in t_all]
        # gradient exchange that did not hide under compute, per rank: main-stream time from "backward enqueued" to "all gradients summed"
        mine = float(np.mean([a.elapsed_time(b) for a, b in comm_marks])) if comm_marks else 0.0
        e_all = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        torch.distributed.all_gather(e_all, torch.tensor([mine], device=dev, dtype=torch.float64))
        exposed_ms = [float(t.item()) for t in e_all]
        # the replicas must stay IDENTICAL (every rank applies the same summed gradients): per-tensor checksums, largest difference to rank 0
        chk = torch.stack([p.detach().double().sum() for p in pt_params + mlp_params] + [p.detach().double().abs().sum() for p in pt_params])
        c_all = [torch.zeros_like(chk) for _ in range(world)]
        torch.distributed.all_gather(c_all, chk)
        replica_spread = float(max((c - c_all[0]).abs().max() for c in c_all))
    if rank == 0:
        from pointnerf_amd.fused import FusedRender
        rays_total = rays_rank * world * args.steps
        rows = float(np.mean([s["n_neighbor_rows"] for s in stats])); smp = float(np.mean([s["n_valid_samples"] for s in stats]))
        headline = args.config == "lego"
        name = ("rays/sec (render only, %d products per multiply-add, supplementary)" % args.inference_products) if args.render_only else "rays/sec (render+bwd)"
        out = {"metric": name + (" NeRF-synth lego 800^2, K=8, 128 samp/ray" if headline else " %s, K=%d, %d samp/ray" % (cfg_name, opt.K, opt.SR)),
               "value": rays_total / dt, "unit": "rays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "median_ms_per_step": median_ms,
               "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
               "dtype": "f32 in / out / accumulate; forward GEMM operands as 2 x f16 planes (22-bit) on the f16 MFMA, 3 products per multiply-add; input-gradient GEMMs: f16 leading product + "
                        "e4m3 cross terms (csrc/mixq.h), 2 product equivalents; weight-gradient GEMM operands as "
                        + ("one f16 plane each" if args.wgrad_planes == 1 else "2 x f16 planes each, 3 products (--wgrad-planes 2: fp32-class)"), "data": "synthetic",
               "neighbor_rows_per_s": rows * world * args.steps / dt, "valid_samples_per_s": smp * world * args.steps / dt,
               "config": {"workload": "BASELINE.json %s, %d neural points, K=%d, SR=%d, D=%d, %s, fwd+loss+bwd+Adam, grid cached"
                                      % (cfg_name, n_points, opt.K, opt.SR, opt.z_depth_dim,
                                         ("configs[2]: one global batch of %d rays/step split contiguously over %d GPUs (%d rays/GPU/step)" % (args.rays, world, rays_rank)) if strong
                                         else ("%d rays/GPU/step" % rays_rank) + (" (weak-scaling variant: global batch %d)" % (rays_rank * world) if world > 1 else "")),
                          "rays_per_gpu_per_step": rays_rank, "global_batch_rays": rays_rank * world,
                          "wgrad_planes": args.wgrad_planes, "cross_terms": dict(zip(("bits", "where_mask"), ops.cross_terms_state())), "collective_selftest": selftest,
                          "parity_note": ("the synthetic Barn shell puts up to ~70 points in a 0.009 cell, beyond P = 11: the reference switches to a wall-clock-seeded "
                                          "reservoir there (parity undefined); the HIP path and the oracle both keep the first P by index, so parity on this "
                                          "configuration is HIP-vs-oracle truncation only (pointnerf_amd/config.py barn_opt)") if args.config == "barn" else None,
                          "parallelism": "ray-shard dp%d, point cloud + MLP replicated" % world, "point_grad_exchange": ("none" if not dist_on else ("zero1 reduce-scatter" if zero1 else ("sparse touched rows" if sparse else "dense all-reduce, one bucket, overlapped"))), "world_size": world, "ms_per_step_by_rank": per_rank_ms, "ms_allreduce_exposed_by_rank": exposed_ms, "replica_param_checksum_spread": replica_spread,
                          "valid_samples_per_step": smp, "neighbor_rows_per_step": rows,
                          "rays_hit_per_step": float(np.mean([s["rays_hit"] for s in stats])), "final_loss": float(loss.item()),
                          "device_allocs_in_timed_region": int(extra_allocs), "setup_steps": 2,
                          "saved_activation_bytes_per_step": int(ops.L.lib().pnerf_agg_saved_bytes(int(smp), int(opt.K))),
                          "arena_budget_bytes": ops.arena_budget_bytes(),
                          "backward_ray_chunks": None if getattr(FusedRender, "last_chunks", None) is None else
                          {"rays_per_chunk": FusedRender.last_chunks[0], "rays": FusedRender.last_chunks[1]},
                          "arithmetic": "f32 inputs / outputs / accumulation throughout; every GEMM of the forward (aggregator and colour MLP) and of the colour "
                                        "MLP's input-gradient chain runs on v_mfma_f32_32x32x16_f16 with every f32 operand carried as two f16 planes "
                                        "(x = h + m to 2^-22) and three products per multiply-add (h*h + h*m + m*h), f32 accumulate: sigma/RGB within "
                                        "1.1e-6 of the f32 oracle at this configuration (bar 1e-4); the aggregator's four input-gradient GEMMs run the leading "
                                        "product h*h on f16 and the two cross terms on e4m3 factors (v_mfma_scale_f32_32x32x64_f8f6f4, csrc/mixq.h: 1.3e-6 rms of "
                                        "sum |terms| per dot product; the LeakyReLU masks come from the fp32-class forward); the weight-gradient GEMMs (aggregator and colour layers) "
                                        "(sums over millions of rows) stream the saved inputs and the output gradients as ONE f16 plane each, rounded to nearest "
                                        "(one product, f32 accumulate; error budget: csrc/backward.hip k_wgrad_f16, tests/test_split_f16_cpu.py; measured "
                                        "against float64: tests/test_gpu_bench_config.py)", **extra}}
        if prof is not None:
            per = {k: {"ms_per_launch": ms / max(n, 1), "launches": n, "ms_per_step": ms / args.steps} for k, (ms, n) in prof.items() if n > 0}
            # algorithmic work per step of the three dominant kernels
            alg_flop = {"agg_forward": rows * FLOP_ROW_FWD, "agg_backward": rows * FLOP_ROW_DGRAD, "wgrad": rows * FLOP_ROW_WGRAD + smp * FLOP_SAMPLE_WGRAD,
                        "color_forward": smp * FLOP_SAMPLE_FWD}
            alg_byte = {"agg_forward": rows * BYTES_ROW_FWD, "agg_backward": rows * BYTES_ROW_BWD, "wgrad": rows * BYTES_ROW_WGRAD + smp * BYTES_SAMPLE_WGRAD}
            # SURVEY 8d's own figures (activations on chip, nothing saved): gather Nv x 172 forward, gradient scatter 2 x Nv x 168 backward; the
            # weight-gradient GEMMs stream nothing in that model (the reference's cuBLAS wgrad reads the activations it kept in HBM)
            survey_byte = {"agg_forward": rows * 172.0, "agg_backward": rows * 2 * 168.0}
            if args.wgrad_planes == 2:
                # --wgrad-planes 2: every saved operand leaves as two planes (X0 whole: 288 columns), the output gradients as two, and the
                # weight-gradient GEMMs stream each (dY plane, X plane) pair of their three products
                xcols = 288 + 256 + 288 + 256
                alg_byte = {"agg_forward": rows * (168 + 2 * 2 * xcols + 2 * 2 * 256 + 16 + 4 + 96),
                            "agg_backward": rows * (2 * 2 * 256 + 96 + 20 + 128 + 128 + 2 * 2 * 4 * 256),
                            "wgrad": 3 * (rows * (2 * xcols + 2 * 4 * 256) + smp * BYTES_SAMPLE_WGRAD)}
            traffic = {}
            tf = os.path.join(ROOT, "profiles", "traffic.json")       # PMC-derived HBM bytes per step (separate rocprofv3 --pmc passes)
            if os.path.exists(tf):
                traffic = json.load(open(tf))

            src = traffic.get("_source")
            traffic_source = TRAFFIC_SOURCE if not src else TRAFFIC_SOURCE + "; collected at commit %s with `%s`" % (src.get("commit"), src.get("command"))
            copy_gbs = measured_copy_gbs(dev)
            mfma_meas = {"random_operands": ops.mfma_rate_tflops(2), "constant_operands": ops.mfma_rate_tflops(1)}

            ct_bits, ct_mask = ops.cross_terms_state()
            products = {"agg_forward": MIX_PRODUCTS_FWD if ct_mask & 2 else float(F16_PRODUCTS), "agg_backward": MIX_PRODUCTS_BWD if ct_mask & 4 else float(F16_PRODUCTS)}
            if args.wgrad_planes == 2:
                products = {"agg_forward": float(F16_PRODUCTS), "agg_backward": float(F16_PRODUCTS)}

            def mfma_entry(k):      # SURVEY 8d's algorithmic flops against the dense f16 MFMA peak; the executed f16-product equivalents beside it
                t = per[k]["ms_per_step"] * 1e-3
                npr = products.get(k, float(F16_PRODUCTS))
                return {"bound": "mfma", "kernel": k, "achieved": alg_flop[k] / t / 1e12, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": alg_flop[k] / t / 1e12 / PEAK_F16_MFMA_TFLOPS,
                        "achieved_executed": npr * alg_flop[k] / t / 1e12,
                        "frac_executed": npr * alg_flop[k] / t / 1e12 / PEAK_F16_MFMA_TFLOPS,
                        "peak_measured": mfma_meas["random_operands"], "peak_measured_constant_operands": mfma_meas["constant_operands"],
                        "frac_executed_of_measured": npr * alg_flop[k] / t / 1e12 / mfma_meas["random_operands"],
                        "peak_measured_note": "register-resident v_mfma_f32_32x32x16_f16 on all CUs, timed in this run (pnerf_debug_mfma_rate): with pseudo-random f16 "
                                              "operands (they toggle like a GEMM's fragments) and with one constant -- the power management holds the clock down "
                                              "when the pipe's inputs switch, so the nominal peak is not reachable on real data",
                        "frac_note": "frac = SURVEY 8d algorithmic flops / dense f16 MFMA peak; frac_executed = executed f16-product equivalents per algorithmic "
                                     "multiply-add (f16x3.h: three f16 products; mixq.h: one f16 product + two e4m3 products at half the pipe cycles = 2) / the same "
                                     "peak = matrix-pipe utilisation in f16 terms",
                        "traffic": traffic.get(k), "traffic_source": traffic_source,
                        "f16_products_per_multiply_add": npr,
                        "cross_terms": "e4m3 (csrc/mixq.h)" if npr < 3 else "f16 (csrc/f16x3.h)",
                        "algorithmic_flop_per_step": alg_flop[k], "ms_per_step": per[k]["ms_per_step"]}

            def hbm_entry(k):
                t = per[k]["ms_per_step"] * 1e-3
                return {"bound": "hbm", "kernel": k, "achieved": alg_byte[k] / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": alg_byte[k] / t / 1e9 / PEAK_HBM_GBS,
                        "peak_measured": copy_gbs, "frac_of_measured": alg_byte[k] / t / 1e9 / copy_gbs,
                        "peak_measured_note": "device-to-device copy of 1 GiB (read + write counted), timed in this run",
                        "traffic": traffic.get(k), "traffic_source": traffic_source, "algorithmic_bytes_per_step": alg_byte[k],
                        "survey_algorithmic_bytes_per_step": survey_byte.get(k),
                        "traffic_over_survey_algorithmic": (traffic[k] / survey_byte[k]) if isinstance(traffic.get(k), (int, float)) and survey_byte.get(k) else None,
                        "ms_per_step": per[k]["ms_per_step"]}
            heavy = [k for k in ("agg_forward", "agg_backward", "wgrad") if k in per]
            if heavy:
                dom = max(heavy, key=lambda k: per[k]["ms_per_step"])
                # the weight-gradient GEMM is HBM-bound by construction (2 KB per row and layer against 0.35 us of MFMA per 16 rows); the two
                # tile kernels are priced against the matrix pipe AND carry their HBM figure
                out["roofline"] = hbm_entry(dom) if dom == "wgrad" else mfma_entry(dom)
                out["roofline_mfma"] = mfma_entry(max([k for k in heavy if k != "wgrad"], key=lambda k: per[k]["ms_per_step"])) if len(heavy) > 1 or dom != "wgrad" else None
                out["roofline_hbm"] = {k: hbm_entry(k) for k in heavy}
            out["kernels"] = per
            # everything of a step that is NOT a libpnerf_hip.so kernel (HIP events around every library launch): the ATen glue of the loss /
            # output dict, memsets, launch gaps and the one host synchronisation
            out["ms_outside_library_kernels"] = dt / args.steps * 1e3 - sum(v["ms_per_step"] for v in per.values())
            for k in alg_flop:
                if k in per:
                    out["kernels"][k]["tflops"] = alg_flop[k] / (per[k]["ms_per_step"] * 1e-3) / 1e12
        if not dist_on and args.cpu_rays > 0 and not args.render_only and args.config in ("lego", "chair"):
            try:
                out["cpu_baseline"] = best_cpu_baseline(opt, n_points, args.cpu_rays, min(os.cpu_count() or 1, 32), points_fn, rays_fn)
                if out["cpu_baseline"].get("kind") == "port":
                    out["cpu_baseline"]["note"] = ("kind 'port': /root/reference does not exist on a GPU box (the checkout cannot travel), so the oracle's restatement is "
                                                   "timed here; the reference's own modules on host cores (kind 'reference': its query kernels compiled for the host + "
                                                   "PointAggregator.forward + ray_march + backward) are timed in the authoring container by `bench.py --cpu-baseline-only` "
                                                   "-> profiles/r06_cpu_baseline_reference.json")
            except Exception as e:       # the checker failing must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
