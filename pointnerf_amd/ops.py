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
    mm = torch.stack([xyz.min(dim=0)[0], xyz.max(dim=0)[0]]).cpu()
    rmin = torch.as_tensor(opt.ranges[:3], dtype=torch.float32)
    rmax = torch.as_tensor(opt.ranges[3:], dtype=torch.float32)
    mn, mx = torch.maximum(mm[0], rmin), torch.minimum(mm[1], rmax)
    pad = torch.as_tensor(scaled_vsize * np.asarray(opt.kernel_size) / 2, dtype=torch.float32)
    mn, mx = mn - pad, mx + pad
    vdim = (mx - mn).numpy() / vsize64
    scaled_vdim = np.ceil(vdim / vscale).astype(np.int32)
    ranges = torch.cat([mn, mx]).numpy().astype(np.float32)
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
