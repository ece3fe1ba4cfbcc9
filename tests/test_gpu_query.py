"""GPU parity of the HIP query path (grid build + probe + neighbor query) through the C ABI,
against the CPU oracle: neighbor indices, sample locations and ray masks BIT-EXACT."""
import numpy as np
import pytest
import torch

from pointnerf_amd import config, scenes
from oracle import pyref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(seed, n, radius=0.06, size=10, **ov):
    kw = dict(K=8, SR=16, P=12, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    kw.update(ov)
    opt = config.lego_opt(**kw)
    xyz = torch.from_numpy(scenes.chair_points(n, seed=seed, radius=radius))
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=17.0 * seed, x0=400 - size // 2, y0=400 - size // 2, size=size))
    return opt, xyz, inp


def _native_op(opt, xyz, inp, hp):
    from pointnerf_amd.point_query import woord_query_grid_point_index
    raypos, _ = pyref.ray_samples(inp["campos"], inp["raydir"], opt.z_depth_dim, float(inp["near"].min()), float(inp["far"].max()))
    R, D = raypos.shape[1:3]
    t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt, device=DEV)
    out = woord_query_grid_point_index(
        inp["pixel_idx"].to(DEV).to(torch.int32), raypos.to(DEV), xyz[None].to(DEV),
        torch.tensor([xyz.shape[0]], dtype=torch.int32, device=DEV), t(opt.kernel_size, torch.int32), t(opt.query_size, torch.int32),
        opt.SR, opt.K, R, D, t(hp["scaled_vdim"], torch.int32), opt.max_o, opt.P, np.float32(hp["radius"]),
        t(hp["ranges"], torch.float32), t(hp["scaled_vsize"], torch.float32), 1024, opt.NN)
    return [o.cpu() for o in out]


def _assert_same(q, pidx, loc, mask):
    assert tuple(pidx.shape) == tuple(q["sample_pidx"].shape), (pidx.shape, q["sample_pidx"].shape)
    assert pidx.dtype == torch.int32 and mask.dtype == torch.int8 and loc.dtype == torch.float32
    assert torch.equal(mask, q["ray_mask"])
    assert torch.equal(pidx, q["sample_pidx"])
    assert torch.equal(loc, q["sample_loc_w"])          # bit-exact floats


@pytest.mark.parametrize("seed,n,K,SR,P,size", [(0, 1200, 8, 16, 12, 10), (1, 700, 4, 8, 12, 9), (2, 3000, 8, 32, 20, 12),
                                                (3, 300, 1, 4, 12, 7), (4, 2000, 6, 24, 16, 10), (5, 2500, 12, 20, 24, 11),
                                                (6, 4000, 16, 128, 30, 8)])
def test_native_op_bit_exact(seed, n, K, SR, P, size):
    opt, xyz, inp = _scene(seed, n, K=K, SR=SR, P=P, size=size)
    q = pyref.query(opt, xyz, inp)
    assert q["info"]["ovf_P"] == 0
    _assert_same(q, *_native_op(opt, xyz, inp, q["hp"]))


def test_native_op_edge_cases():
    # all rays miss
    opt, xyz, inp = _scene(5, 500)
    inp["raydir"] = inp["raydir"] * torch.tensor([1.0, 1.0, -1.0])
    q = pyref.query(opt, xyz, inp)
    pidx, loc, mask = _native_op(opt, xyz, inp, q["hp"])
    assert pidx.shape == (1, 0, opt.SR, opt.K) and loc.shape == (1, 0, opt.SR, 3) and int(mask.sum()) == 0
    # single point / tight ranges (points outside the grid) / kernel 5 + no dilation / radius test off / D not multiple of 64
    for kw, n in [(dict(), 1), (dict(ranges=[-0.05, -0.05, -0.05, 0.05, 0.05, 0.05]), 400),
                  (dict(kernel_size=[5, 5, 5], query_size=[1, 1, 1]), 1500), (dict(radius_limit_scale=0), 800),
                  (dict(z_depth_dim=333), 900)]:
        opt, xyz, inp = _scene(7, n, **kw)
        q = pyref.query(opt, xyz, inp)
        _assert_same(q, *_native_op(opt, xyz, inp, q["hp"]))


def test_querier_config1_chair():
    """BASELINE.json configs[0] through lighting_fast_querier.query_points (fused ray generation)."""
    from pointnerf_amd.point_query import lighting_fast_querier
    opt = config.chair_opt()
    xyz = torch.from_numpy(scenes.chair_points())
    inp = pyref.to_torch_inputs(scenes.block_rays())
    q = pyref.query(opt, xyz, inp, nthreads=4)
    qr = lighting_fast_querier(torch.device(DEV), opt)
    d = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    xyz_d = xyz.to(DEV)
    for _ in range(2):      # second call hits the grid cache
        pidx, loc_p, loc_w, dirs, mask, vsize, ranges = qr.query_points(
            d["pixel_idx"].to(torch.int32), None, xyz_d[None], torch.tensor([xyz.shape[0]], dtype=torch.int32, device=DEV),
            800, 800, inp["intrinsic"].numpy()[0], 2.0, 6.0, d["raydir"], d["campos"], d["camrotc2w"])
        _assert_same(q, pidx.cpu(), loc_w.cpu(), mask.cpu())
        assert pidx.shape == (1, 3282, 32, 4)
        assert torch.equal(dirs.cpu(), q["sample_ray_dirs"])
        assert (loc_p.cpu() - q["sample_loc"]).abs().max() < 1e-5
        assert np.allclose(ranges, q["hp"]["ranges"]) and list(vsize) == list(opt.vsize)
    gi = qr.last_grid_info
    assert gi["n_occ"] == q["info"]["n_occ"] and gi["max_cnt"] >= q["info"]["max_cnt"] and gi["n_in_grid"] == 8192
    # dense counters are consistent with the compacted view
    c = qr.last_dense["counters"].cpu().tolist()
    assert c[1] == 3282 and c[2] == q["info"]["n_sel"] and c[3] == q["info"]["n_neigh"]
    vl = qr.last_dense["valid_list"][:c[0]].cpu()
    assert torch.all(vl[1:] > vl[:-1])
    assert c[0] == int((qr.last_dense["sample_nn"] > 0).sum())


def test_lego_scale_properties_and_subsample_parity():
    """configs[1] size (2M points): determinism, cache idempotence, radius property on the device,
    and bit-exact parity of a 1024-ray subsample against the oracle."""
    from pointnerf_amd.point_query import lighting_fast_querier, clear_grid_cache
    opt = config.bench_lego_opt()
    xyz = torch.from_numpy(scenes.lego_points())
    inp = pyref.to_torch_inputs(scenes.random_rays(3, 16384))
    qr = lighting_fast_querier(torch.device(DEV), opt)
    xyz_d = xyz.to(DEV)
    run = lambda: qr.query_dense(xyz_d[None], xyz.shape[0], 2.0, 6.0, inp["raydir"].to(DEV), inp["campos"].to(DEV))
    a = {k: v.clone() for k, v in run().items()}
    clear_grid_cache()
    b = run()
    for k in ("sample_pidx", "sample_loc", "sample_nn", "ray_hit"):
        assert torch.equal(a[k], b[k]), k
    gi = qr.last_grid_info
    assert not gi["overflow_max_o"], gi
    pidx, loc = a["sample_pidx"].long(), a["sample_loc"]
    valid = pidx >= 0
    d = xyz_d[pidx.clamp(min=0)] - loc[:, :, None, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    r = np.float32(opt.radius_limit_scale * opt.vsize[0])
    assert bool((d2[valid] <= float(r * r) * (1 + 1e-6)).all())
    assert int(valid.sum()) == int(a["counters"][3]) and int(a["ray_hit"].sum()) == int(a["counters"][1])
    # subsample parity
    sub = {k: v for k, v in inp.items()}
    sub["raydir"] = inp["raydir"][:, :1024].contiguous()
    q = pyref.query(opt, xyz, sub, nthreads=8)
    hit = a["ray_hit"][:1024].cpu() > 0
    assert torch.equal(hit.to(torch.int8)[None], q["ray_mask"])
    assert torch.equal(a["sample_pidx"][:1024].cpu()[hit][None], q["sample_pidx"])
    assert torch.equal(a["sample_loc"][:1024].cpu()[hit][None], q["sample_loc_w"])


def _jitter_parity(opt, xyz, inp, dev):
    """training-mode jitter (point_query.py:81): with the uniforms the kernel draws (pnerf_debug_uniform) fed to the oracle's
    restatement of near_far_linear_ray_generation (pinned bit-exactly against the reference, tests/test_oracle_golden.py), the
    selected samples and neighbor indices are identical"""
    from pointnerf_amd import ops
    from pointnerf_amd.point_query import lighting_fast_querier
    qr = lighting_fast_querier(torch.device(dev), opt)
    xyz_d = xyz.to(dev)
    R, D = inp["raydir"].shape[1], opt.z_depth_dim
    dense = qr.query_dense(xyz_d[None], xyz.shape[0], 2.0, 6.0, inp["raydir"].to(dev), inp["campos"].to(dev))
    u = ops.jitter_uniforms(qr.last_seed, R, D, xyz_d.device).cpu()
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and 0.45 < float(u.mean()) < 0.55
    q = pyref.query(opt, xyz, inp, nthreads=8, jitter=0.3, uniforms=u)
    hit = dense["ray_hit"].cpu() > 0
    assert torch.equal(hit.to(torch.int8)[None], q["ray_mask"])
    assert torch.equal(dense["sample_loc"].cpu()[hit][None], q["sample_loc_w"])
    assert torch.equal(dense["sample_pidx"].cpu()[hit][None], q["sample_pidx"])
    # a second call draws new numbers
    dense2 = qr.query_dense(xyz_d[None], xyz.shape[0], 2.0, 6.0, inp["raydir"].to(dev), inp["campos"].to(dev))
    assert not torch.equal(dense["sample_loc"], dense2["sample_loc"])
    return int(hit.sum())


def test_jitter_mode_is_bit_defined():
    opt = config.chair_opt(is_train=1)
    n = _jitter_parity(opt, torch.from_numpy(scenes.chair_points()), pyref.to_torch_inputs(scenes.block_rays()), DEV)
    assert n > 100


def test_jitter_mode_is_bit_defined_at_the_bench_config():
    """the mode bench.py times: is_train = 1 at configs[1] (2 M points, D = 400, SR = 128, K = 8) on a 1024-ray subsample"""
    opt = config.bench_lego_opt(is_train=1)
    d = scenes.random_rays(3, 16384)
    d["raydir"] = d["raydir"][:, :1024]
    n = _jitter_parity(opt, torch.from_numpy(scenes.lego_points()), pyref.to_torch_inputs(d), DEV)
    assert n > 100

def test_grid_cache_never_serves_a_recycled_address():
    """Regression: the grid cache is keyed on the xyz storage address + version.  Two clouds of equal N allocated one
    after the other (the first one freed) used to collide when the caching allocator recycled the address."""
    from pointnerf_amd.point_query import lighting_fast_querier
    opt, _, inp = _scene(3, 600)
    qr = lighting_fast_querier(torch.device(DEV), opt)
    rd, cp = inp["raydir"].to(DEV), inp["campos"].to(DEV)
    for seed in (11, 12, 13, 14):
        xyz = torch.from_numpy(scenes.chair_points(600, seed=seed, radius=0.06))
        q = pyref.query(opt, xyz, inp)
        xd = xyz.to(DEV)
        dense = qr.query_dense(xd[None], 600, 2.0, 6.0, rd, cp)
        hit = dense["ray_hit"].cpu() > 0
        assert torch.equal(dense["sample_pidx"].cpu()[hit][None], q["sample_pidx"]), seed
        del xd, dense


def test_f16_mfma_fragment_layout_and_subnormal_inputs():
    """one v_mfma_f32_32x32x16_f16 through the C ABI: the layout csrc/f16x3.h assumes, and subnormal f16 inputs are NOT flushed"""
    import ctypes
    import numpy as np
    import mfma_case
    from pointnerf_amd import _lib as L
    a, b, D, D_flushed = mfma_case.build()
    da, db = torch.from_numpy(a.view(np.int16)).cuda(), torch.from_numpy(b.view(np.int16)).cuda()
    out = torch.zeros(64, 16, device="cuda")
    L.check(L.lib().pnerf_debug_mfma_f16(ctypes.c_void_p(da.data_ptr()), ctypes.c_void_p(db.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnerf_debug_mfma_f16")
    torch.cuda.synchronize()
    got = mfma_case.unpack(out.cpu().numpy())
    err, err_if_flushed = np.abs(got - D).max(), np.abs(D_flushed - D).max()
    print("max |D - exact| = %.3e (a flushing pipe would give %.3e)" % (err, err_if_flushed))
    assert err <= 1e-6 * np.abs(D).max() and err < 0.01 * err_if_flushed


def test_two_plane_split_is_bit_exact():
    """pn_split2 (v_cvt_pkrtz_f16_f32 + v_fma_mix{lo,hi}_f16) against the numpy restatement of tests/test_split_f16_cpu.py, bit for bit"""
    import ctypes
    import numpy as np
    import mfma_case
    from pointnerf_amd import _lib as L
    x = mfma_case.split_inputs()
    for sat in (0, 1):
        xs = x if sat else np.clip(x, -60000, 60000)
        dx = torch.from_numpy(xs).cuda()
        h, m = torch.zeros(x.size, dtype=torch.int16, device="cuda"), torch.zeros(x.size, dtype=torch.int16, device="cuda")
        L.check(L.lib().pnerf_debug_split(ctypes.c_void_p(dx.data_ptr()), x.size, ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(m.data_ptr()), sat,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnerf_debug_split")
        torch.cuda.synchronize()
        eh, em = mfma_case.split_expected(xs, sat)
        assert np.array_equal(h.cpu().numpy().view(np.uint16), eh.view(np.uint16))
        assert np.array_equal(m.cpu().numpy().view(np.uint16), em.view(np.uint16))


def test_points_minmax_equals_torch_reductions():
    """pnerf_points_minmax (a1: the min / max pass of get_hyperparameters, point_query.py:51-52) against torch.min / torch.max, bit for bit: mixed
    signs, denormals, +-0, one point, 2 M points"""
    import ctypes
    from pointnerf_amd import _lib as L
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(9)
    for n in (1, 7, 70001, 2_000_000):
        x = (torch.randn(n, 3, generator=g) * torch.tensor([1e-3, 1.0, 1e4])).to(dev)
        if n > 5:
            x[3, 0] = 0.0; x[4, 0] = -0.0; x[5, 1] = 1e-41; x[2, 2] = -1e-41
        out = torch.full((6,), float("nan"), device=dev)
        L.check(L.lib().pnerf_points_minmax(ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(out.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnerf_points_minmax")
        want = torch.cat([x.min(dim=0)[0], x.max(dim=0)[0]])
        assert bool((out.cpu() == want.cpu()).all()), (n, out, want)          # (value equality: -0.0 == 0.0)
