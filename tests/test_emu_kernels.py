"""The HIP kernels executed on the HOST (tools/emu: same sources, every GPU thread a fiber, MFMA / shuffles / barriers
emulated) against the CPU oracle.  These run in the CPU suite: a kernel indexing, layout or synchronisation bug fails here,
before any GPU time is spent; the -m gpu tests remain the parity tests proper (the emulator's MFMA accumulation order and
transcendental functions are the host's, so the tolerances here are the same bars, not tighter ones)."""
import numpy as np
import pytest
import torch

import test_gpu_backward as TB
import test_gpu_render as TR
from cases import build_case
from emu_util import emu_backend
from pointnerf_amd import config, scenes
from oracle import pyref


@pytest.fixture(autouse=True)
def _emu(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(TB, "DEV", "cpu")
    with emu_backend():
        yield


@pytest.fixture(autouse=True)
def _poison(monkeypatch):
    """the saved-activation arena starts as NaN bit patterns (as in the -m gpu tests): nothing unwritten may reach a result"""
    from pointnerf_amd import ops
    orig = ops.Arena.take

    def take(self, nbytes, device):
        t = orig(self, nbytes, device)
        t.fill_(0xFF)
        return t

    monkeypatch.setattr(ops.Arena, "take", take)


def _tiny_case(K, SR, size, n=1200, seed=5):
    opt = config.lego_opt(K=K, SR=SR, P=24, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    xyz = torch.from_numpy(scenes.chair_points(n, seed=seed, radius=0.06))
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, 32, seed).items()}
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=55.0, x0=400 - size // 2, y0=400 - size // 2, size=size))
    mlp = pyref.init_mlp_params(opt, seed=3, bias_scale=0.1)
    return opt, xyz, attrs, inp, mlp


def test_emulated_forward_matches_oracle():
    TR._compare(*build_case("small_k4"))


@pytest.mark.parametrize("K", [256, 272, 288])
def test_emulated_mixed_tile_gemm_matches_the_numpy_restatement(K):
    """csrc/mixq.h: LDS row format, packed image, block scales, unit schedule -- the emulated kernel equals the numpy restatement of the
    format to fp32 accumulation noise, and both are within 4e-6 of sum |terms| of the exact product"""
    import ctypes
    import mix_case
    from emu_util import emu_lib
    for big in (False, True):
        x, w = mix_case.build(K, big=big)
        out = np.zeros((64, 256), np.float32)
        img = np.zeros(emu_lib().pnerf_mlp_packed_bytes(), np.uint8)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert emu_lib().pnerf_debug_mix_gemm(P(w), K, P(x), P(img), P(out), None) == 0
        ref = mix_case.restate(x, w)
        ex, sab = mix_case.exact(x, w)
        assert np.abs(out - ref).max() <= 2e-6 * np.abs(ref).max(), np.abs(out - ref).max()
        e = (out - ex) / sab
        print("K", K, "big", big, "rms err / sum|terms| %.2e  max %.2e" % (np.sqrt((e ** 2).mean()), np.abs(e).max()))
        assert np.abs(e).max() <= 1e-5 and np.sqrt((e ** 2).mean()) <= 2e-6


def test_emulated_forward_with_e4m3_cross_terms_stays_inside_the_bar():
    """the optional forward modes of csrc/mixq.h (pnerf_set_cross_terms_where bits 0 / 1) on the emulator: sigma / RGB / ray colour within 1e-4 of the
    oracle (the default, f16 cross terms in the forward, is what every other test of this file runs)"""
    from pointnerf_amd import ops
    old = ops.set_cross_terms(8, where=7)
    try:
        errs = TR._compare(*build_case("small_k4"))
    finally:
        ops.set_cross_terms(8, where=old[1])
    assert 2e-6 < errs["decoded"] <= 1e-4          # (the mode is really on: f16 cross terms give < 1e-6)


def test_emulated_f16_mfma_layout_and_subnormals():
    import ctypes
    import mfma_case
    from emu_util import emu_lib
    a, b, D, _ = mfma_case.build()
    out = np.zeros((64, 16), np.float32)
    rc = emu_lib().pnerf_debug_mfma_f16(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), None)
    assert rc == 0
    assert np.abs(mfma_case.unpack(out) - D).max() <= 1e-6 * np.abs(D).max()


def test_emulated_split_matches_numpy():
    import ctypes
    import mfma_case
    from emu_util import emu_lib
    x = mfma_case.split_inputs()
    for sat in (0, 1):
        xs = x if sat else np.clip(x, -60000, 60000)
        h, m = np.zeros(x.size, np.float16), np.zeros(x.size, np.float16)
        assert emu_lib().pnerf_debug_split(xs.ctypes.data_as(ctypes.c_void_p), x.size, h.ctypes.data_as(ctypes.c_void_p), m.ctypes.data_as(ctypes.c_void_p), sat, None) == 0
        eh, em = mfma_case.split_expected(xs, sat)
        assert np.array_equal(h.view(np.uint16), eh.view(np.uint16)) and np.array_equal(m.view(np.uint16), em.view(np.uint16))


# K % 4 == 0 runs three sample classes (K, K/2, K/4 rows per sample), other K one
@pytest.mark.parametrize("K,SR,size", [(8, 12, 5), (3, 10, 4), (12, 8, 4), (1, 6, 5), (16, 6, 3)])
def test_emulated_forward_and_backward_match_oracle(K, SR, size):
    case = _tiny_case(K, SR, size)
    TR._compare(*case)
    TB._run(*case)


def test_emulated_level1_chain(monkeypatch):
    """NeuralPoints.forward -> PointAggregator.forward -> ray_march as separate modules (the reference's forward body) with
    autograd through them, on per-slot gathered arrays (tests/test_gpu_level1.py on a reduced case)"""
    import cases
    import test_gpu_level1 as T1
    monkeypatch.setattr(T1, "DEV", "cpu")
    monkeypatch.setitem(cases.CASES, "small_k4", (dict(K=4, SR=10, P=12, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3]), 900, 5, 1))
    T1.test_level1_chain_matches_oracle_and_fused("small_k4")


def test_emulated_jitter_is_bit_defined():
    """the jittered ray sampling (sequential running sum per ray, counter RNG) against the oracle fed the same uniforms"""
    import test_gpu_query as TQ
    opt, xyz, attrs, inp, mlp = _tiny_case(8, 12, 5)
    opt.is_train = 1
    assert TQ._jitter_parity(opt, xyz, inp, "cpu") > 5


def test_emulated_backward_by_ray_chunks(monkeypatch):
    """arena budget exceeded -> inference forward + chunked recompute backward (fused._backward_in_chunks) equals the one-pass step"""
    opt, xyz, attrs, inp, mlp = _tiny_case(4, 8, 4)
    step, R = TB._chunked_equals_one_pass(opt, xyz, attrs, inp, mlp, "cpu", 0.0055, monkeypatch)
    assert step < R


@pytest.mark.parametrize("seed,n,K,SR,size,kw", [
    (0, 1200, 8, 16, 6, dict(P=12)),
    (6, 4000, 16, 70, 4, dict(P=30)),                                                # > 64 slots per ray: two wave passes
    (2, 3000, 8, 12, 5, dict(P=32, vsize=[0.02, 0.02, 0.02])),                       # fat cells: the shell splits into several LDS chunks
    (7, 400, 8, 12, 5, dict(ranges=[-0.05, -0.05, -0.05, 0.05, 0.05, 0.05])),        # samples in the grid's border cells
    (7, 1500, 8, 12, 5, dict(kernel_size=[1, 1, 1])),                                # one layer: the own cell only
    (7, 1500, 8, 12, 5, dict(kernel_size=[5, 5, 5], query_size=[1, 1, 1])),          # three layers: the thread-per-sample kernel
    (3, 2500, 3, 12, 5, dict(P=40)),                                                 # P beyond the tile rows: the thread-per-sample kernel
])
def test_emulated_neighbor_query_bit_exact(monkeypatch, seed, n, K, SR, size, kw):
    """k_probe + k_neighbors_tiles (LDS-staged candidate tiles) / k_neighbors: indices, sample positions and masks equal the oracle's"""
    import test_gpu_query as TQ
    monkeypatch.setattr(TQ, "DEV", "cpu")
    opt, xyz, inp = TQ._scene(seed, n, K=K, SR=SR, size=size, **kw)
    q = pyref.query(opt, xyz, inp)
    assert int((q["sample_pidx"] >= 0).sum()) > 50
    TQ._assert_same(q, *TQ._native_op(opt, xyz, inp, q["hp"]))


def test_emulated_fused_zero_one_loss_matches_the_torch_chain():
    """ops.ZeroOneConf (one fused pass forward, one backward) against the chain it replaces: gather with the -1 -> point 0 rule,
    gradient_clamp, clamp(eps, 1 - eps), log + log(1 - .), sum; value and the gradient on points_conf"""
    from pointnerf_amd import ops
    from pointnerf_amd.neural_points_volumetric_model import gradient_clamp
    g = torch.Generator().manual_seed(3)
    N, M, eps = 500, 6000, 1e-3
    conf = torch.rand(1, N, 1, generator=g) * 1.2 - 0.1                 # some below 1e-4 / eps, some above 1 - eps and above 1
    conf[0, :5, 0] = torch.tensor([0.0, 1e-4, 1e-3, 1 - 1e-3, 1.0])     # the clamp bounds themselves
    pidx = torch.randint(-1, N, (M,), generator=g, dtype=torch.int32)
    pidx[torch.rand(M, generator=g) < 0.4] = -1                         # empty slots: the point-0 flood
    a = conf.clone().requires_grad_(True)
    cc = gradient_clamp(a.reshape(-1)[pidx.long().clamp(min=0)])
    v = cc.clamp(eps, 1 - eps)
    ref = (torch.log(v) + torch.log(1 - v)).sum()
    (ref * 0.37).backward()
    b = conf.clone().requires_grad_(True)
    got = ops.zero_one_conf_sum(b, pidx, eps)
    (got * 0.37).backward()
    assert abs(float(got) - float(ref)) <= 2e-5 * abs(float(ref))
    assert torch.allclose(b.grad, a.grad, rtol=2e-5, atol=1e-5 * float(a.grad.abs().max()))


def test_emulated_two_plane_weight_gradients():
    """pnerf_set_wgrad_planes(2): the saved operands' residual planes, the three-product weight-gradient launches and the whole-X0 path of the
    fused step, on the emulator: same bars as the default mode, and the MLP gradients of the two modes agree to the one-plane rounding."""
    from pointnerf_amd import ops
    case = _tiny_case(8, 12, 5)
    gm_o, gp_o, probe = TB._oracle_grads(*case)
    one, _, _, _ = TB._hip_grads(*case, probe)
    old = ops.set_wgrad_planes(2)
    try:
        assert old == 1
        two, gp2, _, _ = TB._hip_grads(*case, probe)
    finally:
        assert ops.set_wgrad_planes(old) == 2
    for k in gm_o:
        TB._check(k, two[k], gm_o[k])
    for k in gp_o:
        TB._check(k, gp2[k], gp_o[k])
    for k in gm_o:       # one plane against two planes: the 2^-11 roundings of a few hundred rows, far below the parity bar
        scale = max(float(two[k].abs().max()), 1e-12)
        assert float((one[k] - two[k]).abs().max()) <= 2e-4 * scale, k


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_emulated_extract_2d_and_query_embedding(tag):
    """csrc/embed2d.hip on the emulator: MvsPointsModel.query_embedding (projection, in-image / z-buffer masks, bilinear sampling of the
    image + feature pyramid, per-view directions) against the oracle and against the reference's own functions (refembed.npz)."""
    from embed_case import run_case
    with emu_backend():
        err = run_case(tag, "cpu")
    assert max(err.values()) <= 1e-5, err


def test_emulated_step_with_a_capacity_above_the_actual_count(monkeypatch):
    """The render step enqueued with the ARENA'S CAPACITY as its bound instead of the step's own number of valid samples (round 4: a training
    step is enqueued before its counters have reached the host): every kernel must take the actual counts from the device.  Same outputs and
    gradients as the exact call, on a NaN-poisoned arena, for a capacity 1.6 x + 13 above the count."""
    from pointnerf_amd import ops
    case = _tiny_case(8, 12, 5)
    exact = TB._model_grads(*case, "cpu")
    rf, rb = ops.render_forward, ops.render_backward
    seen = []

    def fwd(cam, pts, packed, flat, raydir, dense, R, SR, K, n_valid, train):
        seen.append(n_valid)
        return rf(cam, pts, packed, flat, raydir, dense, R, SR, K, int(n_valid * 1.6) + 13, train)

    def bwd(cam, pts, packed, flat, raydir, dense, R, SR, K, n_valid, *a, **k):
        return rb(cam, pts, packed, flat, raydir, dense, R, SR, K, int(n_valid * 1.6) + 13, *a, **k)

    monkeypatch.setattr(ops, "render_forward", fwd)
    monkeypatch.setattr(ops, "render_backward", bwd)
    # (FusedRender sizes nothing else from env["n_valid"]; the arena block is taken for the larger bound inside render_forward)
    loose = TB._model_grads(*case, "cpu")
    assert seen and seen[0] > 0
    # the weight-gradient GEMMs split their rows over the workgroups by the host bound: another bound, another summation order (last bits);
    # everything else is bit-identical
    for k in exact:
        if k.startswith("points_") or k.startswith("alpha_branch"):
            assert torch.equal(exact[k], loose[k]), k
        else:
            err = float((exact[k] - loose[k]).abs().max()) / max(float(exact[k].abs().max()), 1e-12)
            assert err <= 2e-6, (k, err)


@pytest.mark.parametrize("K,SR,size", [(8, 12, 5), (12, 8, 4)])
def test_emulated_fused_losses_match_the_aten_chains(K, SR, size):
    """the zero-one regulariser inside the render node (its conf gradient on the backward's own conf atomics, the empty slots in closed form)
    and the colour loss over the dense ray colours against the ATen chains they replace: loss and every gradient of the step; K = 8 runs
    the one-pass front, K = 12 the two-pass one"""
    TB.fused_losses_equivalence(*_tiny_case(K, SR, size), "cpu")


def test_emulated_extract_2d_edge_cases():
    """csrc/embed2d.hip: no points; every point outside every image (rows of zeros, masks false); feature layers only (no colour map:
    colors is None, the mask comes from the first feature map of a view); a one-channel map sampled exactly at its corner texels."""
    import types
    from shell_fakes import embed_inputs
    from pointnerf_amd.mvs_points_model import MvsPointsModel
    inp = embed_inputs(seed=3, n=64)
    m = MvsPointsModel(types.SimpleNamespace(depth_occ=0, ref_vid=0, shading_feature_mlp_layer0=0))
    common = (inp["intrinsics"], inp["c2ws"], inp["w2cs"])
    f, c = m.extract_2d(inp["img_feats"], [0, 1], [0, 1], *common, inp["cam_xyz"][:, :0], inp["HD"], inp["WD"])
    assert f.shape == (1, 0, 16) and c.shape == (1, 0, 6)
    far = inp["cam_xyz"].clone()
    far[..., 0] += 100.0                                            # far to the side: outside every image
    for occ in (0, 1):
        m.args.depth_occ = occ
        f, c, mask = m.extract_2d(inp["img_feats"], [0, 1, 2], [0, 1, 2, 3], *common, far, inp["HD"], inp["WD"], return_mask=True)
        assert float(f.abs().max()) == 0.0 and float(c.abs().max()) == 0.0 and not bool(mask.any())
    m.args.depth_occ = 0
    f, c, mask = m.extract_2d(inp["img_feats"], [1], [2, 3], *common, inp["cam_xyz"], inp["HD"], inp["WD"], cam_vid=0, return_mask=True)
    assert c is None and f.shape == (1, 64, 48) and mask.shape == (1, 64)
    assert torch.equal(mask[0], f[0].abs().sum(-1) > 0)
    # corner texels: a point that projects exactly onto pixel (0, 0) / (WD - 1, HD - 1) of the current camera reads that texel unblended
    HD, WD = inp["HD"], inp["WD"]
    K = inp["intrinsics"][0, 0]
    ramp = (torch.arange(HD * WD, dtype=torch.float32).reshape(1, 1, HD, WD)).repeat(3, 1, 1, 1)
    z = 2.0
    pts = torch.tensor([[[(0 - K[0, 2]) / K[0, 0] * z, (0 - K[1, 2]) / K[1, 1] * z, z],
                         [((WD - 1) - K[0, 2]) / K[0, 0] * z, ((HD - 1) - K[1, 2]) / K[1, 1] * z, z]]], dtype=torch.float32)
    f, c = m.extract_2d([inp["img_feats"][0], ramp], [0], [1], *common, pts, HD, WD, cam_vid=0)
    assert torch.allclose(f[0, :, 0], torch.tensor([0.0, float(HD * WD - 1)]), atol=2e-3), f


def test_emulated_training_forward_without_backward_releases_its_arena():
    """A training-mode forward that is never back-propagated (an evaluation under enabled gradients) must not keep its activation arena alive:
    no reference cycle through the autograd node (the node keeps decoded / weight / opacity for its backward -- they must be
    non-differentiable outputs, or tensor -> grad_fn -> node -> tensor holds the arena until a cycle collection; round 4 met it as an
    out-of-memory error after seventeen 16 GB forwards)."""
    import gc
    import weakref
    from pointnerf_amd.neural_points import NeuralPoints
    from pointnerf_amd.point_aggregators import PointAggregator
    from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
    opt, xyz, attrs, inp, mlp = _tiny_case(8, 12, 5)
    agg = PointAggregator(opt)
    agg.load_state_dict(mlp)
    agg.flatten_()
    npnt = NeuralPoints(32, xyz.shape[0], opt, torch.device("cpu"))
    npnt.set_points(xyz, attrs["points_embeding"], points_color=attrs["points_color"], points_dir=attrs["points_dir"], points_conf=attrs["points_conf"], parameter=True)
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    gc.collect()
    gc.disable()
    try:
        for flags in ((False, False), (True, True)):
            model.fused_zero_one, model.fused_color_loss = flags
            out = model(**inp)
            colour = out["_dense_color"][0] if "_dense_color" in out else out["coarse_raycolor"]
            node = colour.grad_fn
            while node is not None and type(node).__name__ != "FusedRenderBackward":
                node = node.next_functions[0][0] if node.next_functions else None
            assert node is not None
            ref = weakref.ref(node)
            del out, colour, node
            assert ref() is None, "the render node survives its outputs: a reference cycle (fused flags %s)" % (flags,)
    finally:
        gc.enable()
