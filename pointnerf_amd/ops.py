"""Thin host wrappers: torch tensors (device memory, streams) -> C-ABI calls of libpnerf_hip.so.

PyTorch is plumbing here: it owns HBM allocations and the stream; every computation on the hot
path happens inside the library.  All functions require CUDA(=HIP) tensors and raise otherwise.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib as L


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "pointnerf_amd ops need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def host_array(t):
    """The small tensors of a step that the host needs as numbers (camera position / rotation, background, near / far, Rw2c: 3 .. 9 floats)
    as a numpy array.  A device tensor is read back ONCE per (tensor object, version): the copy is cached on the tensor object itself and is
    valid as long as the tensor's in-place version counter has not moved (a new tensor -- the reference's ``set_input`` makes one per batch --
    is read back again).  Every read-back is a host synchronisation with the device idle behind it: a step made eight of them (near, far,
    camera position twice, rotation, background, Rw2c, counters) where one is needed."""
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    if not t.is_cuda:
        return t.detach().numpy()
    # Only step INPUTS are cached: a Parameter (Rw2c, when it is learnable) is written through .data / optimizer kernels without a version
    # bump, and an inference-mode tensor has no version counter at all (t._version raises): both are read back every time.
    cacheable = not isinstance(t, torch.nn.Parameter) and not t.requires_grad
    ver = None
    if cacheable:
        try:
            ver = t._version
        except Exception:
            cacheable = False
    if cacheable:
        c = getattr(t, "_pnerf_host", None)
        if c is not None and c[0] == ver and c[2] == t.data_ptr():      # (data_ptr: t.data = ... / set_() re-home a tensor without a version bump)
            return c[1]
    arr = t.detach().cpu().numpy()
    arr.setflags(write=False)                                            # callers share the cached copy
    if cacheable:
        try:
            t._pnerf_host = (ver, arr, t.data_ptr())
        except Exception:
            pass
    return arr


def _need_cuda(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("pointnerf_amd: %s must be a device tensor (this path has no CPU implementation)" % name)


# ------------------------------------------------------------------------------------------ grid
def mid_depths(D, near, far):
    """The D mid-point depths of near_far_linear_ray_generation with jitter=0
    (models/rendering/diff_ray_marching.py:369-384), computed on the host with the same fp32 op
    sequence (linspace, lerp, diff, cumsum, midpoint) so that samples are bit-identical."""
    t = torch.linspace(0, 1, D + 1).view(1, -1)
    t = near * (1 - t) + far * t
    seg = (t[..., 1:] - t[..., :-1]) * (1 + 0.0 * (torch.zeros(1, 1, D) - 0.5))
    end = torch.cumsum(seg, dim=2)
    end = near + torch.cat([torch.zeros(1, 1, 1), end], dim=2)
    mid = (end[:, :, :-1] + end[:, :, 1:]) / 2
    return mid.reshape(-1).contiguous(), seg.reshape(-1).contiguous()


def grid_hyperparameters(opt, xyz):
    """lighting_fast_querier.get_hyperparameters (models/neural_points/point_query.py:47-71) and the
    constants of :35-42, read from ``opt`` at call time.  One device->host sync (the min/max),
    amortised by the grid cache."""
    vsize64 = np.asarray(opt.vsize, dtype=np.float64)
    vscale = np.asarray(opt.vscale, dtype=np.int32)
    scaled_vsize = (np.asarray(opt.vsize) * vscale).astype(np.float32)
    radius = np.asarray(opt.radius_limit_scale * max(opt.vsize[0], opt.vsize[1])).astype(np.float32)
    _need_cuda(xyz, "xyz")
    xyz = xyz.contiguous()
    mm6 = torch.empty(6, dtype=torch.float32, device=xyz.device)
    L.check(L.lib().pnerf_points_minmax(_ptr(xyz), int(xyz.shape[0]), _ptr(mm6), _stream()), "pnerf_points_minmax")
    mm = mm6.cpu().numpy().reshape(2, 3)                  # (the one device -> host read of a grid rebuild; fp32 arithmetic below, like the reference's torch ops)
    rmin = np.asarray(opt.ranges[:3], dtype=np.float32)
    rmax = np.asarray(opt.ranges[3:], dtype=np.float32)
    mn, mx = np.maximum(mm[0], rmin), np.minimum(mm[1], rmax)
    pad = (scaled_vsize * np.asarray(opt.kernel_size) / 2).astype(np.float32)
    mn, mx = (mn - pad).astype(np.float32), (mx + pad).astype(np.float32)
    vdim = (mx - mn).astype(np.float32) / vsize64
    scaled_vdim = np.ceil(vdim / vscale).astype(np.int32)
    ranges = np.concatenate([mn, mx]).astype(np.float32)
    return ranges, scaled_vsize, scaled_vdim, float(radius)


def make_grid_params(ranges, scaled_vsize, scaled_vdim, kernel_size, query_size, P, max_o, radius):
    gp = L.GridParams()
    gp.ranges[:] = [float(x) for x in ranges]
    gp.vsize[:] = [float(x) for x in scaled_vsize]
    gp.vdim[:] = [int(x) for x in scaled_vdim]
    gp.kernel_size[:] = [int(x) for x in kernel_size]
    gp.query_size[:] = [int(x) for x in query_size]
    gp.P, gp.max_o, gp.radius = int(P), int(max_o), float(radius)
    return gp


class VoxelGrid:
    """A built grid: params + the device workspace that holds it."""

    def __init__(self, gp, ws, n_points):
        self.gp, self.ws, self.n_points = gp, ws, n_points
        self._info = None

    def info(self):
        """dict(n_in_grid, n_occ, max_cnt, cell0, first_idx); synchronises the stream once."""
        if self._info is None:
            buf = (ctypes.c_int32 * L.GI_LEN)()
            L.check(L.lib().pnerf_grid_info(_ptr(self.ws), buf, _stream()), "pnerf_grid_info")
            self._info = dict(n_in_grid=buf[0], n_occ=buf[1], max_cnt=buf[2], cell0=buf[3], first_idx=buf[4],
                              overflow_max_o=buf[1] > self.gp.max_o, overflow_P=buf[2] > self.gp.P)
        return self._info


def build_grid(gp, xyz):
    """xyz [N,3] f32 device tensor -> VoxelGrid (enqueues only)."""
    _need_cuda(xyz, "xyz")
    xyz = xyz.detach().reshape(-1, 3).contiguous().float()
    n = xyz.shape[0]
    lib = L.lib()
    nbytes = lib.pnerf_grid_workspace_bytes(ctypes.byref(gp), n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
    L.check(lib.pnerf_grid_build(ctypes.byref(gp), _ptr(xyz), n, _ptr(ws), nbytes, _stream()), "pnerf_grid_build")
    return VoxelGrid(gp, ws, n)


# ------------------------------------------------------------------------------------------ query
def query_dense(grid, R, D, SR, K, raypos=None, campos=None, raydir=None, mid=None, near=0.0, far=0.0,
                jitter=0.0, seed=0):
    """pnerf_query: dense-over-R outputs, no host sync.  Returns a dict of device tensors."""
    dev = grid.ws.device
    lib = L.lib()
    loc = torch.empty(R, SR, 3, dtype=torch.float32, device=dev)
    pidx = torch.empty(R, SR, K, dtype=torch.int32, device=dev)
    nn = torch.empty(R, SR, dtype=torch.int32, device=dev)
    hit = torch.empty(R, dtype=torch.int32, device=dev)
    vlist = torch.empty(max(R * SR, 1), dtype=torch.int32, device=dev)
    counters = torch.empty(8, dtype=torch.int32, device=dev)
    nws = lib.pnerf_query_workspace_bytes(R, SR)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    cam = None
    if raypos is None:
        cam = (ctypes.c_float * 3)(*[float(x) for x in campos])
        _need_cuda(raydir, "raydir"); _need_cuda(mid, "mid")
    L.check(lib.pnerf_query(ctypes.byref(grid.gp), _ptr(grid.ws), _ptr(raypos), cam, _ptr(raydir), _ptr(mid),
                            float(near), float(far), float(jitter), int(seed) & (2 ** 64 - 1), R, D, SR, K,
                            _ptr(loc), _ptr(pidx), _ptr(nn), _ptr(hit), _ptr(vlist), _ptr(counters),
                            _ptr(ws), nws, _stream()), "pnerf_query")
    return dict(sample_loc=loc, sample_pidx=pidx, sample_nn=nn, ray_hit=hit, valid_list=vlist, counters=counters)


def jitter_uniforms(seed, R, D, device):
    """The uniforms U[r, d] the jittered query draws for ``seed`` (pnerf_debug_uniform): [1, R, D] f32 device tensor."""
    out = torch.empty(1, R, D, dtype=torch.float32, device=device)
    L.check(L.lib().pnerf_debug_uniform(int(seed) & (2 ** 64 - 1), 0, R * D, _ptr(out), _stream()), "pnerf_debug_uniform")
    return out


# ------------------------------------------------------------------------------------------ MLP + renderer
def mlp_layout():
    """name -> (offset, shape) of the flat parameter vector, reference state_dict order."""
    offs = (ctypes.c_int64 * (L.MLP_NTENSORS + 1))()
    L.check(L.lib().pnerf_mlp_layout(32, offs), "pnerf_mlp_layout")
    names = ["block1.0", "block1.2", "block3.0", "block3.2", "alpha_branch.0",
             "color_branch.0", "color_branch.2", "color_branch.4", "color_branch.6"]
    shapes = [(256, 284), (256, 256), (256, 263), (256, 256), (1, 256), (128, 280), (128, 128), (128, 128), (3, 128)]
    out, i = {}, 0
    for n, shp in zip(names, shapes):
        out[n + ".weight"] = (offs[i], shp); out[n + ".bias"] = (offs[i + 1], (shp[0],)); i += 2
    return out, offs[L.MLP_NTENSORS]


def flatten_mlp(state, device):
    """dict of reference-named tensors -> flat fp32 device vector."""
    lay, total = mlp_layout()
    flat = torch.empty(total, dtype=torch.float32, device=device)
    for k, (o, shp) in lay.items():
        flat[o:o + int(np.prod(shp))] = state[k].detach().reshape(-1).to(device=device, dtype=torch.float32)
    return flat


def pack_mlp(flat, out=None):
    """Re-pack the flat parameters into MFMA fragment order (must be called after every weight update)."""
    _need_cuda(flat, "mlp parameters")
    lib = L.lib()
    if out is None:
        out = torch.empty(lib.pnerf_mlp_packed_bytes(), dtype=torch.uint8, device=flat.device)
    L.check(lib.pnerf_mlp_pack(_ptr(flat), _ptr(out), _stream()), "pnerf_mlp_pack")
    return out


def make_camera(campos, camrot, vsize_z, raydist_mode_unit=1, bg=None, rw2c=None):
    c = L.Camera()
    c.campos[:] = [float(x) for x in np.asarray(campos, dtype=np.float64).reshape(-1)[:3]]
    c.camrot[:] = [float(x) for x in np.asarray(camrot, dtype=np.float64).reshape(-1)[:9]]
    r = np.eye(3) if rw2c is None else np.asarray(rw2c, dtype=np.float64)
    c.rw2c[:] = [float(x) for x in r.reshape(-1)[:9]]
    c.vsize_z = float(vsize_z)
    c.raydist_mode_unit = int(raydist_mode_unit)
    if bg is None:
        c.has_bg = 0
        c.bg[:] = [0.0, 0.0, 0.0]
    else:
        c.has_bg = 1
        c.bg[:] = [float(x) for x in np.asarray(bg, dtype=np.float64).reshape(-1)[:3]]
    return c


def make_points(xyz, emb, conf, pdir, color):
    for n, t in (("xyz", xyz), ("points_embeding", emb), ("points_conf", conf), ("points_dir", pdir), ("points_color", color)):
        _need_cuda(t, n)
        assert t.is_contiguous() and t.dtype == torch.float32, n
    p = L.Points()
    p.xyz, p.embedding, p.conf, p.dir, p.color = xyz.data_ptr(), emb.data_ptr(), conf.data_ptr(), pdir.data_ptr(), color.data_ptr()
    p.n, p.feat_dim = int(xyz.reshape(-1, 3).shape[0]), int(emb.shape[-1])
    return p


class Arena:
    """Grow-only device scratch for the saved activations of one in-flight training forward.  The arena is tens of GB
    at bench size and its exact size changes every step with the number of valid samples; letting torch's caching
    allocator see a new size each step costs a hipMalloc/hipFree pair of that size (measured: ~1.9 s per step)."""

    def __init__(self):
        self.free, self.headroom = [], 1.15

    def take(self, nbytes, device):
        best = None
        for t in self.free:
            if t.numel() >= nbytes and t.device == device and (best is None or t.numel() < best.numel()):
                best = t
        if best is not None:
            self.free.remove(best)
            return best
        self.free = [t for t in self.free if t.device != device]      # drop too-small blocks before growing
        return torch.empty(int(nbytes * self.headroom) + 256, dtype=torch.uint8, device=device)

    def give(self, t):
        if t is not None:
            self.free.append(t)

    def capacity_samples(self, K, device):
        """the number of valid samples the largest free block can hold the saved activations of (0: no free block yet): the capacity a
        render step can be ENQUEUED with before the step's own count has reached the host (NeuralPointsRayMarching.render_dense)"""
        best = max((t.numel() for t in self.free if t.device == torch.device(device)), default=0)
        if best <= 0:
            return 0
        key = (best, int(K))
        hit = getattr(self, "_cap_cache", {}).get(key)
        if hit is not None:
            return hit
        fn = L.lib().pnerf_agg_saved_bytes
        lo, hi = 0, best // 512
        while lo < hi:                          # largest n with saved_bytes(n, K) <= best (monotone)
            mid = (lo + hi + 1) // 2
            if fn(mid, int(K)) <= best:
                lo = mid
            else:
                hi = mid - 1
        self._cap_cache = {key: lo}
        return lo

    def reserve(self, nbytes, device):
        """make sure one free block of at least ``nbytes`` exists (a caller that knows its largest step sizes the arena once,
        instead of paying a ~2 s hipFree + hipMalloc of tens of GB when a later step is 15 % larger than every earlier one)"""
        if not any(t.numel() >= nbytes and t.device == device for t in self.free):
            self.give(self.take(nbytes, device))


ARENA = Arena()


def arena_budget_bytes():
    """Upper bound for the saved-activation arena of ONE render step (5.9 KB per neighbor row incl. the per-sample areas: 44 GB at the bench configuration).
    A training step whose rows would need more (Barn-scale clouds at K = 12, large ray batches) does not fail or swap: its forward runs
    in inference mode and its backward re-runs the forward chunk of rays by chunk of rays, each chunk within the budget
    (fused.FusedRender).  PNERF_ARENA_BUDGET_GB overrides the default of 160 GB (of the 288 GB of an MI355X)."""
    import os
    return int(float(os.environ.get("PNERF_ARENA_BUDGET_GB", "160")) * (1 << 30))


def compact_valid(sample_nn):
    """work list of the samples with neighbors: (valid_list [n] i32, counters [8] i32) from sample_nn [R,SR] (pnerf_compact_valid)"""
    lib = L.lib()
    n = sample_nn.numel()
    dev = sample_nn.device
    vlist = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    counters = torch.empty(8, dtype=torch.int32, device=dev)
    nws = lib.pnerf_compact_workspace_bytes(n)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    L.check(lib.pnerf_compact_valid(_ptr(sample_nn), n, _ptr(vlist), _ptr(counters), _ptr(ws), nws, _stream()), "pnerf_compact_valid")
    return vlist, counters


def touched_flags(pidx, n_points):
    """[n_points] int32 0/1 flags of the points that occur in the int32 neighbor table ``pidx`` (row 0 also when a slot is empty):
    pnerf_touched_flags, one pass over the table, no int64 temporaries"""
    _need_cuda(pidx, "pidx")
    if pidx.dtype != torch.int32 or not pidx.is_contiguous():
        raise ValueError("touched_flags: a contiguous int32 neighbor table is expected")
    flags = torch.empty(max(int(n_points), 0), dtype=torch.int32, device=pidx.device)
    L.check(L.lib().pnerf_touched_flags(_ptr(pidx), pidx.numel(), int(n_points), _ptr(flags), _stream()), "pnerf_touched_flags")
    return flags


def reserve_pool(nbytes, device):
    """Pre-size torch's caching allocator for the step's variable-size tensors (everything indexed by the number of rays
    that hit the cloud: compacted weights / indices / confidences, the loss temporaries and their gradients).  Their sizes
    change with every batch, so without a pool the allocator keeps meeting requests no cached block fits and calls
    hipMalloc in the middle of training steps (measured: 3 per step, each a device-wide stall, 15 ms per step on a
    freshly booted box).  One block of the worst-case size, allocated and released once, is split and re-merged by the
    allocator from then on."""
    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    del t


def render_forward(cam, pts, packed, flat, raydir, dense, R, SR, K, n_valid, train):
    """pnerf_render_forward.  n_valid = host copy of dense['counters'][0] (capacity of the scratch).
    Returns dict(decoded, weight, ray_color, opacity, bg_trans, blend_w, saved)."""
    dev = raydir.device
    lib = L.lib()
    f32 = dict(dtype=torch.float32, device=dev)
    decoded = torch.empty(R, SR, 4, **f32); weight = torch.empty(R, SR, K, **f32)
    ray_color = torch.empty(R, 3, **f32); opacity = torch.empty(R, SR, **f32)
    bg_trans = torch.empty(R, **f32); blend_w = torch.empty(R, SR, **f32)
    saved = ws = None
    if train:
        saved = ARENA.take(lib.pnerf_agg_saved_bytes(n_valid, K), dev)
        nws = 0
    else:
        nws = lib.pnerf_agg_workspace_bytes(n_valid, K)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    L.check(lib.pnerf_render_forward(ctypes.byref(cam), ctypes.byref(pts), _ptr(packed), _ptr(flat), _ptr(raydir),
                                     _ptr(dense["sample_loc"]), _ptr(dense["sample_pidx"]), _ptr(dense["sample_nn"]),
                                     _ptr(dense["valid_list"]), _ptr(dense["counters"]), R, SR, K,
                                     _ptr(decoded), _ptr(weight), _ptr(ray_color), _ptr(opacity), _ptr(bg_trans), _ptr(blend_w),
                                     _ptr(saved), n_valid, _ptr(ws), nws, _stream()), "pnerf_render_forward")
    return dict(decoded=decoded, weight=weight, ray_color=ray_color, opacity=opacity, bg_trans=bg_trans,
                blend_w=blend_w, saved=saved)


def render_backward(cam, pts, packed, flat, raydir, dense, R, SR, K, n_valid, fwd, grad_ray_color, grad_flat, grads, ready_event=None, zero_one=None):
    """pnerf_render_backward: accumulates into grad_flat (MLP) and grads = dict(points_embeding=..., points_conf=...,
    points_dir=..., points_color=...) (device tensors, same shapes as the parameters).  ``zero_one`` = (scale [1] f32 device tensor, eps): the
    conf gradient of the zero-one regulariser over the hit rays' neighbor table is added by the same call (pnerf_point_grads.zero_one_gscale;
    ``dense`` must be the query's own output: its counters [1] and [3] count the empty slots)."""
    lib = L.lib()
    pg = L.PointGrads()
    pg.embedding, pg.conf = grads["points_embeding"].data_ptr(), grads["points_conf"].data_ptr()
    pg.dir, pg.color = grads["points_dir"].data_ptr(), grads["points_color"].data_ptr()
    if zero_one is not None:
        pg.zero_one_gscale, pg.zero_one_eps = zero_one[0].data_ptr(), float(zero_one[1])
    if ready_event is not None:          # a torch.cuda.Event that has been recorded once (so that its hipEvent_t exists)
        pg.ready_event = ready_event.cuda_event
    nws = lib.pnerf_render_backward_workspace_bytes(R, SR)
    ws = torch.empty(nws, dtype=torch.uint8, device=raydir.device)
    g = grad_ray_color.contiguous().float()
    L.check(lib.pnerf_render_backward(ctypes.byref(cam), ctypes.byref(pts), _ptr(packed), _ptr(flat), _ptr(raydir),
                                      _ptr(dense["sample_loc"]), _ptr(dense["sample_pidx"]), _ptr(dense["sample_nn"]),
                                      _ptr(dense["valid_list"]), _ptr(dense["counters"]), R, SR, K, n_valid,
                                      _ptr(fwd["decoded"]), _ptr(fwd["weight"]), _ptr(fwd["opacity"]), _ptr(g),
                                      _ptr(fwd["saved"]), _ptr(grad_flat), ctypes.byref(pg), _ptr(ws), nws, _stream()),
            "pnerf_render_backward")


# ------------------------------------------------------------------------------------------ gather (autograd)
class GatherRows(torch.autograd.Function):
    """rows = src[max(idx, 0)]  (NeuralPoints.forward's index_select, neural_points.py:706-717) with the
    scatter-add backward, both in libpnerf_hip.so.  src [N,W] f32, idx [...] i32 -> [..., W]."""

    @staticmethod
    def forward(ctx, src, idx):
        _need_cuda(src, "src")
        s2 = src.detach().reshape(-1, src.shape[-1]).contiguous().float()
        i2 = idx.reshape(-1).contiguous().to(torch.int32)
        out = torch.empty(i2.numel(), s2.shape[1], dtype=torch.float32, device=src.device)
        L.check(L.lib().pnerf_gather_rows(_ptr(s2), s2.shape[0], s2.shape[1], _ptr(i2), i2.numel(), _ptr(out), _stream()),
                "pnerf_gather_rows")
        ctx.save_for_backward(i2)
        ctx.src_shape = tuple(src.shape)
        return out.view(tuple(idx.shape) + (s2.shape[1],))

    @staticmethod
    def backward(ctx, g):
        (i2,) = ctx.saved_tensors
        w = ctx.src_shape[-1]
        gs = torch.zeros(ctx.src_shape, dtype=torch.float32, device=g.device)
        g2 = g.reshape(-1, w).contiguous().float()
        n_src = gs.numel() // w
        L.check(L.lib().pnerf_scatter_add_rows(_ptr(g2), _ptr(i2), i2.numel(), w, _ptr(gs), n_src, _stream()),
                "pnerf_scatter_add_rows")
        return gs, None


def gather_rows(src, idx):
    return GatherRows.apply(src, idx)


class ZeroOneConf(torch.autograd.Function):
    """sum over the neighbor slots of  log(v) + log(1 - v),  v = clamp(gradient_clamp(conf[max(pidx, 0)], 1e-4, 1), eps, 1 - eps): the
    numerator of the reference's ``loss_zero_one`` on ``conf_coefficient`` (models/base_rendering_model.py:630-641) without materialising
    the [R'', SR, K] tensor -- one HIP pass forward, one backward (pnerf_zero_one_forward / _backward)."""

    @staticmethod
    def forward(ctx, conf, pidx, eps):
        lib = L.lib()
        c = conf.detach().reshape(-1)
        _need_cuda(c, "points_conf")
        idx = pidx.reshape(-1)
        nb = lib.pnerf_zero_one_blocks(idx.numel())
        part = torch.empty(nb, dtype=torch.float32, device=c.device)
        L.check(lib.pnerf_zero_one_forward(_ptr(c), c.numel(), _ptr(idx), idx.numel(), float(eps), _ptr(part), _stream()), "pnerf_zero_one_forward")
        ctx.save_for_backward(c, idx)
        ctx.eps, ctx.shape = float(eps), conf.shape
        return part.sum()

    @staticmethod
    def backward(ctx, g):
        c, idx = ctx.saved_tensors
        grad = torch.zeros_like(c)
        gs = g.detach().reshape(1).to(torch.float32).contiguous()
        L.check(L.lib().pnerf_zero_one_backward(_ptr(c), c.numel(), _ptr(idx), idx.numel(), ctx.eps, _ptr(gs), _ptr(grad), _stream()), "pnerf_zero_one_backward")
        return grad.view(ctx.shape), None, None


def zero_one_conf_sum(conf, pidx, eps):
    return ZeroOneConf.apply(conf, pidx.contiguous(), eps)


class ZeroOneConfRays(torch.autograd.Function):
    """ZeroOneConf over the DENSE neighbor table [R, SR, K] of a query restricted to the rays that hit (ray_hit [R] int32 > 0): the
    reference's conf_coefficient exists for the hit rays only; this form never copies the table to [R'', SR, K]
    (pnerf_zero_one_forward_rays / _backward_rays)."""

    @staticmethod
    def forward(ctx, conf, pidx, ray_hit, eps):
        lib = L.lib()
        c = conf.detach().reshape(-1)
        _need_cuda(c, "points_conf")
        R = int(pidx.shape[0])
        slots = int(pidx.numel() // max(R, 1))
        nb = lib.pnerf_zero_one_blocks(R * 256)
        part = torch.empty(nb, dtype=torch.float32, device=c.device)
        L.check(lib.pnerf_zero_one_forward_rays(_ptr(c), c.numel(), _ptr(pidx), _ptr(ray_hit), R, slots, float(eps), _ptr(part), _stream()), "pnerf_zero_one_forward_rays")
        ctx.save_for_backward(c, pidx, ray_hit)
        ctx.eps, ctx.shape, ctx.dims = float(eps), conf.shape, (R, slots)
        return part.sum()

    @staticmethod
    def backward(ctx, g):
        c, pidx, ray_hit = ctx.saved_tensors
        grad = torch.zeros_like(c)
        gs = g.detach().reshape(1).to(torch.float32).contiguous()
        L.check(L.lib().pnerf_zero_one_backward_rays(_ptr(c), c.numel(), _ptr(pidx), _ptr(ray_hit), ctx.dims[0], ctx.dims[1], ctx.eps, _ptr(gs), _ptr(grad), _stream()),
                "pnerf_zero_one_backward_rays")
        return grad.view(ctx.shape), None, None, None


def zero_one_conf_sum_rays(conf, pidx_dense, ray_hit, eps):
    return ZeroOneConfRays.apply(conf, pidx_dense.contiguous(), ray_hit.contiguous(), eps)


class ColorLossRays(torch.autograd.Function):
    """sum over the rays that hit of (colour - gt)^2 on the DENSE ray colours [R,3] (pnerf_color_loss_forward_rays / _backward_rays): the
    colour term of the training loss without the hit rays' compaction (argsort + index_selects + their scatter-back in the backward)."""

    @staticmethod
    def forward(ctx, ray_color, gt, ray_hit):
        _need_cuda(ray_color, "ray_color")
        lib = L.lib()
        c, g = ray_color.detach().reshape(-1, 3).contiguous().float(), gt.detach().reshape(-1, 3).contiguous().float()
        R = c.shape[0]
        part = torch.empty(lib.pnerf_color_loss_blocks(R), dtype=torch.float32, device=c.device)
        L.check(lib.pnerf_color_loss_forward_rays(_ptr(c), _ptr(g), _ptr(ray_hit), R, _ptr(part), _stream()), "pnerf_color_loss_forward_rays")
        ctx.save_for_backward(c, g, ray_hit)
        ctx.shape = ray_color.shape
        return part.sum()

    @staticmethod
    def backward(ctx, gout):
        c, g, ray_hit = ctx.saved_tensors
        grad = torch.empty_like(c)
        gs = gout.detach().reshape(1).to(torch.float32).contiguous()
        L.check(L.lib().pnerf_color_loss_backward_rays(_ptr(c), _ptr(g), _ptr(ray_hit), c.shape[0], _ptr(gs), _ptr(grad), _stream()),
                "pnerf_color_loss_backward_rays")
        return grad.view(ctx.shape), None, None


def color_loss_sum_rays(ray_color, gt, ray_hit):
    return ColorLossRays.apply(ray_color, gt, ray_hit.contiguous())


def set_inference_products(n):
    """Products per multiply-add of the inference forward: 3 (default, fp32-class accuracy, what the training forward always runs) or
    2 (render / evaluation option: ~1.5x less matrix work, ray colour within ~2e-5 of fp32).  Returns the previous setting."""
    old = L.lib().pnerf_set_inference_products(int(n))
    if old < 0:
        raise ValueError("inference products must be 2 or 3")
    return old


def set_wgrad_planes(n):
    """f16 planes per operand of the weight-gradient GEMMs of the training backward: 1 (default: one plane rounded to nearest, one MFMA
    product) or 2 (both operands as two planes, three products: fp32-class weight gradients, the reference arithmetic of the convergence
    A/B in tests/test_gpu_zz_convergence.py; twice the saved / streamed bytes).  Process-wide; choose it between steps, never between a
    training forward and its backward.  Returns the previous setting."""
    old = L.lib().pnerf_set_wgrad_planes(int(n))
    if old < 0:
        raise ValueError("weight-gradient planes must be 1 or 2")
    ARENA._cap_cache = {}                 # pnerf_agg_saved_bytes depends on the mode: a capacity cached under the other mode is off by ~2x
    return old


def set_cross_terms(bits, where=None):
    """Arithmetic of the cross terms h*m + m*h of the aggregator's tile GEMMs (csrc/mixq.h): 8 = e4m3 factors (default), 16 = f16 factors (csrc/f16x3.h,
    rounds 2-5).  `where` (optional) = which kernels use the e4m3 form: bit 0 inference forward, bit 1 training forward, bit 2 backward (default 4).
    Returns (previous bits, previous mask)."""
    old = L.lib().pnerf_set_cross_terms(int(bits))
    if old < 0:
        raise ValueError("cross terms must be 8 or 16")
    oldw = L.lib().pnerf_set_cross_terms_where(int(where)) if where is not None else None
    if where is not None and oldw < 0:
        raise ValueError("cross-term mask must be 0..7")
    return old, oldw


def cross_terms_state():
    """(bits, mask) currently set: see set_cross_terms"""
    lib = L.lib()
    bits = lib.pnerf_set_cross_terms(8)
    lib.pnerf_set_cross_terms(bits)
    mask = lib.pnerf_set_cross_terms_where(4)
    lib.pnerf_set_cross_terms_where(mask)
    return bits, (mask if bits == 8 else 0)


# ------------------------------------------------------------------------------------------ profiling
def mfma_rate_tflops(mode=2, ms_target=60.0, device=None):
    """TFLOP/s of register-resident v_mfma_f32_32x32x16_f16 on the whole chip (pnerf_debug_mfma_rate; mode 0 zero operands, 1 one constant,
    2 pseudo-random f16: operands that toggle like a GEMM's): the matrix-pipe ceiling bench.py reports beside the nominal 2.5 PFLOP/s.
    The clock needs time to come up from idle and to settle under the power management: launches of ~``ms_target`` are repeated until two
    consecutive ones agree to 3 % (at most eight) and the last one is reported -- the SUSTAINED rate (the first launches of a cold process run at the
    clock's way up from idle, the first ones with toggling operands at a clock the power management has not yet taken back)."""
    import ctypes
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    scratch = torch.zeros(256, dtype=torch.float32, device=dev)
    flop = ctypes.c_double(0.0)
    iters = max(int(ms_target / 1000.0 * 2.4e9 / (2 * 32 * 32)), 64)          # two waves per SIMD x 32 MFMAs of 32 cycles per iteration
    last = 0.0
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().pnerf_debug_mfma_rate(int(mode), int(iters), _ptr(scratch), ctypes.byref(flop), _stream()), "pnerf_debug_mfma_rate")
        e1.record()
        e1.synchronize()
        rate = flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
        settled = last > 0.0 and abs(rate - last) <= 0.03 * last
        last = rate
        if settled:
            break
    return last


def prof_enable(on=True):
    L.lib().pnerf_prof_enable(1 if on else 0)


def prof_collect():
    """{kernel name: (total ms, launches)} since the last collect; synchronises the device."""
    lib = L.lib()
    n = lib.pnerf_prof_kernel_count()
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int64 * n)()
    L.check(lib.pnerf_prof_collect(ms, cnt), "pnerf_prof_collect")
    return {lib.pnerf_prof_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}
