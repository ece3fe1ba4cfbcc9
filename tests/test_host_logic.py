"""CPU tests of the host-side mirror of the reference interface (no kernels involved)."""
import os

import numpy as np
import torch

from emu_util import emu_backend

from pointnerf_amd import config, ops, scenes, dist as pdist
from pointnerf_amd.point_aggregators import PointAggregator
from oracle import pyref


def test_grid_hyperparameters_match_oracle():
    for opt, xyz in [(config.chair_opt(), scenes.chair_points()), (config.lego_opt(), scenes.lego_points(50000)),
                     (config.lego_opt(vscale=[3, 3, 3], kernel_size=[5, 5, 5], ranges=[-0.05] * 3 + [0.05] * 3), scenes.chair_points(500))]:
        x = torch.from_numpy(xyz)
        hp = pyref.grid_hyperparameters(opt, x)
        with emu_backend():            # (the min / max pass is a HIP kernel since round 6: here on the host emulator, tools/emu)
            ranges, svs, svd, radius = ops.grid_hyperparameters(opt, x)
        assert np.array_equal(ranges, hp["ranges"]) and np.array_equal(svs, hp["scaled_vsize"])
        assert np.array_equal(svd, hp["scaled_vdim"]) and radius == hp["radius"]


def test_mid_depths_bit_equal_to_reference_ray_generation():
    fix = np.load(__file__.replace("test_host_logic.py", "golden/raygen_pe.npz"))
    mid, seg = ops.mid_depths(400, 2.0, 6.0)
    assert np.array_equal(mid.numpy(), fix["mid"])           # golden: the reference's near_far_linear_ray_generation
    assert abs(float(seg.sum()) - 4.0) < 1e-4


def test_aggregator_container_has_reference_state_dict():
    opt = config.lego_opt()
    agg = PointAggregator(opt)
    sd = agg.state_dict()
    shapes = pyref.mlp_param_shapes(opt)
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert sum(v.numel() for v in sd.values()) == 341764
    # xavier gains of init_seq: leaky gain for layers followed by LeakyReLU, 1 for the heads; biases zero
    g = np.sqrt(2.0 / (1 + 0.01 ** 2))
    assert abs(float(sd["block1.0.weight"].abs().max()) - g * np.sqrt(6.0 / (284 + 256))) < 2e-3
    assert abs(float(sd["color_branch.6.weight"].abs().max()) - np.sqrt(6.0 / (128 + 3))) < 5e-3
    assert float(sd["block3.2.bias"].abs().max()) == 0.0
    lay, total = ops.mlp_layout()
    assert total == 341764 and list(lay.keys()) == list(sd.keys())
    offs = [o for o, _ in lay.values()]
    assert offs == sorted(offs) and lay["color_branch.6.bias"][0] + 3 == total


def test_arena_reuses_and_grows():
    a = ops.Arena()
    t1 = a.take(1000, torch.device("cpu"))
    assert t1.numel() >= 1000
    a.give(t1)
    t2 = a.take(900, torch.device("cpu"))
    assert t2 is t1
    a.give(t2)
    t3 = a.take(5000, torch.device("cpu"))
    assert t3.numel() >= 5000 and t3 is not t1 and not a.free


def test_shard_slices_partition_the_batch():
    for n, w in [(65536, 8), (1000, 3), (7, 8), (0, 2)]:
        idx = []
        for r in range(w):
            s = pdist.shard_slice(n, r, w)
            idx += list(range(n))[s]
        assert idx == list(range(n))


def test_scene_generators_are_deterministic():
    a, b = scenes.lego_points(20000), scenes.lego_points(20000)
    assert np.array_equal(a, b) and a.shape == (20000, 3) and a.dtype == np.float32
    r1, r2 = scenes.random_rays(3, 128), scenes.random_rays(3, 128)
    assert np.array_equal(r1["raydir"], r2["raydir"]) and r1["raydir"].shape == (1, 128, 3)
    c2w, intr = scenes.synth_camera(30.0)
    assert abs(intr[0, 0] - 1111.111) < 1e-2 and abs(np.linalg.norm(c2w[:3, 3]) - 4.0) < 1e-5


def test_bench_watchdog_leaves_a_record_and_exits():
    """bench.py's guard around the first RCCL contact: a block that does not finish in time ends the process with code 17 and one JSON line on stderr
    (stage, rank, RCCL version, the NCCL_/HSA_ environment, the NCCL_DEBUG hint); a block that does finish is left alone."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog('fast stage', 5, 0, 8): pass\n"
            "with bench.Watchdog('the stuck stage', 1, 3, 8): time.sleep(30)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, NCCL_DEBUG="WARN"))
    assert r.returncode == 17, (r.returncode, r.stderr[-500:])
    rec = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
    assert "the stuck stage" in rec["bench_watchdog"] and rec["rank"] == 3 and rec["world"] == 8 and rec["env"]["NCCL_DEBUG"] == "WARN" and "NCCL_DEBUG=INFO" in rec["hint"]


def test_fill_invalid_on_dense_results_equals_the_scatter_form():
    """neural_points_volumetric_model.py:87-123 two ways: the compacted form (the R'' hit rays scattered back into full-size tensors: the reference's)
    and the dense form a training step with the fused colour loss hands over (per-ray selects on results that are dense already): same tensors."""
    import torch
    from pointnerf_amd.neural_points_volumetric_model import fill_invalid
    g = torch.Generator().manual_seed(3)
    R, SR = 37, 6
    hit = torch.rand(R, generator=g) < 0.6
    ray_color, opacity, bg_trans = torch.rand(R, 3, generator=g), torch.rand(R, SR, generator=g), torch.rand(R, generator=g)
    idx = torch.nonzero(hit).squeeze(1)
    bg = torch.tensor([[0.2, 0.9, 0.4]])
    compact = dict(ray_mask=hit.to(torch.int8)[None], _hit_index=idx, coarse_raycolor=ray_color[idx][None], coarse_point_opacity=opacity[idx][None],
                   coarse_is_background=bg_trans[idx][None, :, None], queried_shading=torch.zeros(1, idx.numel(), 3))
    dense = dict(ray_mask=hit.to(torch.int8)[None], _dense_color=(ray_color, hit.to(torch.int32), int(hit.sum())), _dense_aux=(opacity, bg_trans))
    for bg_ray in (None, torch.rand(1, R, 3, generator=g)):
        a, b = fill_invalid(compact, bg, bg_ray=bg_ray), fill_invalid(dense, bg, bg_ray=bg_ray)
        for k in ("coarse_raycolor", "coarse_point_opacity", "coarse_is_background", "coarse_mask", "queried_shading"):
            assert a[k].shape == b[k].shape and torch.allclose(a[k], b[k], rtol=0, atol=1e-7), (k, bg_ray is not None)
        assert not b["coarse_raycolor"].requires_grad


def test_microbench_figures_do_not_regress_between_rounds():
    """profiles/rNN_microbench.json (tools/gpu_microbench.py on an MI355X, committed once per round): no kernel of the latest round is more than 10 %
    slower than in the round before.  (Round 4 lost 36 % of the stand-alone embedding gather -- the loop vectoriser it had relied on was switched off
    library-wide -- and nobody noticed for two rounds; round 6 restored it with explicit 16-byte accesses.)"""
    import glob
    import json
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_microbench.json")), key=lambda p: int(re.search(r"r(\d+)_", os.path.basename(p)).group(1)))
    assert len(files) >= 2
    new, old = json.load(open(files[-1])), json.load(open(files[-2]))
    checked = 0
    for k, v in new.items():
        if isinstance(v, dict) and "ms" in v and isinstance(old.get(k), dict) and "ms" in old[k]:
            assert v["ms"] <= 1.10 * old[k]["ms"], (k, v["ms"], old[k]["ms"], os.path.basename(files[-1]), os.path.basename(files[-2]))
            checked += 1
    assert checked >= 4
    g = new["gather_rows_emb32"]
    assert g["GBps"] >= 4000.0, g          # the figure DESIGN.md quotes for the stand-alone gather
