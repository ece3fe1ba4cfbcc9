"""Drop-in for the reference's ``models/neural_points/point_query.py`` (the ``wcoord_query=-1`` path).

Same names, argument order, return arity, dtypes and shapes as the reference:
  * ``woord_query_grid_point_index(...)``  -- the native op bound at
    ``models/neural_points/cuda/query_worldcoords.cpp:34-82`` (call site ``point_query.py:86-93``);
  * ``lighting_fast_querier(device, opt).query_points(...)`` -- ``point_query.py:25-98``.
Everything heavy runs in libpnerf_hip.so (pointnerf_amd/csrc/{grid,query}.hip).  Differences in
behaviour, all deliberate and documented in DESIGN.md:
  * the voxel grid is cached while ``xyz`` and the grid parameters are unchanged (the reference rebuilds
    it from all N points on every call, query_worldcoords.cu:308-365);
  * ``opt.kernel_size/query_size`` are read at call time (the reference caches them at construction,
    point_query.py:39-42, and so ignores run/train_ft.py:428's mutation);
  * overflow of ``max_o``/``P`` -- where the reference switches to a wall-clock-seeded curand reservoir --
    keeps the first P points by index / all voxels, and is reported through ``last_grid_info``;
  * K up to 16 is supported (the reference's torch-ext kernel overflows its 8-entry buffer for K>8).
"""
import numpy as np
import torch

from . import ops

_GRID_CACHE = {}
_GRID_CACHE_MAX = 4


def _cache_key(xyz, n_actual, gp_tuple):
    return (xyz.data_ptr(), xyz._version, tuple(xyz.shape), int(n_actual), gp_tuple, xyz.device.index)


def _cache_get(key):
    e = _GRID_CACHE.get(key)
    return None if e is None else e[0]


def _cached_grid(xyz, n_actual, gp_tuple, make_gp):
    """The grid is a function of (xyz contents, parameters).  Identity of the contents = (storage address, version
    counter); every entry keeps a reference to its xyz tensor so that the address cannot be recycled by the
    allocator for a different cloud while the entry lives (observed: two test scenes of equal N sharing an address)."""
    key = _cache_key(xyz, n_actual, gp_tuple)
    g = _cache_get(key)
    if g is None:
        if len(_GRID_CACHE) >= _GRID_CACHE_MAX:
            _GRID_CACHE.pop(next(iter(_GRID_CACHE)))
        g = ops.build_grid(make_gp(), xyz.reshape(-1, 3)[:n_actual])
        _GRID_CACHE[key] = (g, xyz)
    return g


def clear_grid_cache():
    _GRID_CACHE.clear()


def _compact(dense, SR, K):
    """The reference's second ray compaction (query_worldcoords.cu:425-429): keep rays that own at
    least one neighbor.  One host sync (the boolean index), as unavoidable as the reference's."""
    hit = dense["ray_hit"] > 0
    pidx = dense["sample_pidx"][hit].unsqueeze(0)
    loc = dense["sample_loc"][hit].unsqueeze(0)
    return pidx, loc, hit.to(torch.int8).unsqueeze(0)


def woord_query_grid_point_index(pixel_idx_tensor, raypos_tensor, point_xyz_w_tensor, actual_numpoints_tensor,
                                 kernel_size, query_size, SR, K, R, D, scaled_vdim, max_o, P, radius_limit,
                                 ranges, scaled_vsize, kMaxThreadsPerBlock=1024, NN=2):
    """Native-op signature of the reference (B=1).  ``pixel_idx_tensor``, ``kMaxThreadsPerBlock`` and ``NN``
    are accepted and ignored exactly as the reference ignores them."""
    assert point_xyz_w_tensor.shape[0] == 1, "batch size 1 only (every reference script uses batch_size=1)"
    assert raypos_tensor.shape[1] == R and raypos_tensor.shape[2] == D
    as_list = lambda t: [x for x in (t.detach().cpu().tolist() if isinstance(t, torch.Tensor) else list(np.asarray(t).tolist()))]
    ks, qs, vd, rg, vs = as_list(kernel_size), as_list(query_size), as_list(scaled_vdim), as_list(ranges), as_list(scaled_vsize)
    n_actual = int(actual_numpoints_tensor.reshape(-1)[0].item()) if isinstance(actual_numpoints_tensor, torch.Tensor) else int(actual_numpoints_tensor)
    gp_tuple = (tuple(np.float32(rg).tolist()), tuple(np.float32(vs).tolist()), tuple(vd), tuple(ks), tuple(qs), int(P), int(max_o), float(np.float32(radius_limit)))
    make_gp = lambda: ops.make_grid_params(rg, vs, vd, ks, qs, P, max_o, float(np.float32(radius_limit)))
    grid = _cached_grid(point_xyz_w_tensor, n_actual, gp_tuple, make_gp)
    raypos = raypos_tensor.detach().reshape(R, D, 3).contiguous().float()
    dense = ops.query_dense(grid, R, D, SR, K, raypos=raypos)
    pidx, loc, mask = _compact(dense, SR, K)
    return [pidx, loc, mask]


class lighting_fast_querier():
    """``lighting_fast_querier`` of the reference (point_query.py:25-98) on top of libpnerf_hip.so."""

    def __init__(self, device, opt):
        self.device = device if isinstance(device, torch.device) else torch.device(device)
        self.gpu = self.device.index
        self.opt = opt
        self.inverse = getattr(opt, "inverse", 0)
        self.count = 0
        self.last_grid = None
        self.last_dense = None
        self._mid_cache = {}

    def clean_up(self):
        self.last_grid = None
        self.last_dense = None

    @property
    def last_grid_info(self):
        return None if self.last_grid is None else self.last_grid.info()

    def _grid(self, point_xyz_w_tensor, n_actual):
        opt = self.opt
        xyz = point_xyz_w_tensor.reshape(-1, 3)
        opt_key = (tuple(opt.vsize), tuple(opt.vscale), tuple(opt.kernel_size), tuple(opt.query_size), tuple(opt.ranges),
                   int(opt.P), int(opt.max_o), float(opt.radius_limit_scale))
        holder = {}

        def make_gp():
            return holder["gp"]

        # the hyper-parameters depend on min/max of xyz: compute them only on a cache miss
        g = _cache_get(_cache_key(point_xyz_w_tensor, n_actual, opt_key))
        if g is None:
            ranges, svs, svd, radius = ops.grid_hyperparameters(opt, xyz[:n_actual])
            holder["gp"] = ops.make_grid_params(ranges, svs, svd, opt.kernel_size, opt.query_size, opt.P, opt.max_o, radius)
            g = _cached_grid(point_xyz_w_tensor, n_actual, opt_key, make_gp)
        return g

    def query_dense(self, point_xyz_w_tensor, actual_numpoints, near_depth, far_depth, ray_dirs_tensor, cam_pos_tensor):
        """The fused form used by the fast path: dense [R,...] device tensors, no host sync."""
        opt = self.opt
        if getattr(opt, "inverse", 0) > 0:
            raise NotImplementedError("inverse-depth sampling (opt.inverse>0) is outside the hot-path scope (SURVEY.md 2 #7)")
        near_depth, far_depth = float(np.asarray(near_depth).item()), float(np.asarray(far_depth).item())
        grid = self._grid(point_xyz_w_tensor, int(actual_numpoints))
        D = int(opt.z_depth_dim)
        mk = (D, near_depth, far_depth, ray_dirs_tensor.device.index)
        if mk not in self._mid_cache:
            mid, seg = ops.mid_depths(D, near_depth, far_depth)
            self._mid_cache = {mk: (mid.to(ray_dirs_tensor.device), seg.to(ray_dirs_tensor.device))}
        mid, seg = self._mid_cache[mk]
        # point_query.py:81 of the reference; ``opt.ray_jitter`` (ours, default None = that rule) pins it, e.g. to 0 for
        # parity runs of a training step: the jittered depths come from a device RNG and are not reproducible across devices
        jitter = 0.3 if opt.is_train > 0 else 0.0
        if getattr(opt, "ray_jitter", None) is not None:
            jitter = float(opt.ray_jitter)
        raydir = ray_dirs_tensor.detach().reshape(-1, 3).contiguous().float()
        R = raydir.shape[0]
        campos = ops.host_array(cam_pos_tensor).reshape(-1)[:3].tolist() if isinstance(cam_pos_tensor, torch.Tensor) else list(cam_pos_tensor)
        self.count += 1
        self.last_seed = self.count * 0x9E3779B1       # the jitter uniforms of this call are pnerf_debug_uniform(last_seed, ray * D + d)
        dense = ops.query_dense(grid, R, D, int(opt.SR), int(opt.K), campos=campos, raydir=raydir,
                                mid=seg if jitter > 0 else mid, near=near_depth, far=far_depth, jitter=jitter,
                                seed=self.last_seed)
        self.last_grid, self.last_dense = grid, dense
        return dense

    def query_points(self, pixel_idx_tensor, point_xyz_pers_tensor, point_xyz_w_tensor, actual_numpoints_tensor, h, w,
                     intrinsic, near_depth, far_depth, ray_dirs_tensor, cam_pos_tensor, cam_rot_tensor):
        opt = self.opt
        n_actual = int(actual_numpoints_tensor.reshape(-1)[0].item())
        dense = self.query_dense(point_xyz_w_tensor, n_actual, near_depth, far_depth, ray_dirs_tensor, cam_pos_tensor)
        sample_pidx_tensor, sample_loc_w_tensor, ray_mask_tensor = _compact(dense, opt.SR, opt.K)
        sample_ray_dirs_tensor = torch.masked_select(ray_dirs_tensor, ray_mask_tensor[..., None] > 0).reshape(
            ray_dirs_tensor.shape[0], -1, 3)[..., None, :].expand(-1, -1, opt.SR, -1).contiguous()
        gp = self.last_grid.gp
        return sample_pidx_tensor, self.w2pers(sample_loc_w_tensor, cam_rot_tensor, cam_pos_tensor), \
            sample_loc_w_tensor, sample_ray_dirs_tensor, ray_mask_tensor, opt.vsize, np.asarray(list(gp.ranges), dtype=np.float32)

    def w2pers(self, point_xyz_w, camrotc2w, campos):
        """point_query.py:101-108 (plain torch: three small elementwise ops on [1,R'',SR,3])."""
        xyz_w_shift = point_xyz_w - campos[:, None, :]
        xyz_c = torch.sum(xyz_w_shift[..., None, :] * torch.transpose(camrotc2w, 1, 2)[:, None, None, ...], dim=-1)
        z_pers = xyz_c[..., 2]
        x_pers = xyz_c[..., 0] / xyz_c[..., 2]
        y_pers = xyz_c[..., 1] / xyz_c[..., 2]
        return torch.stack([x_pers, y_pers, z_pers], dim=-1)
