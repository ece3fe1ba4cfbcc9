"""Drop-in for the reference's ``models/neural_points/neural_points.py`` class ``NeuralPoints`` (the parts on
the hot path and the point-cloud mutators the run scripts call).  Parameter NAMES are checkpoint keys
(``neural_points.xyz|points_embeding|points_conf|points_dir|points_color|Rw2c``, neural_points.py:243-288) and are
kept verbatim, including the reference's spelling.

  forward(inputs)            -> the reference's 14-tuple (neural_points.py:699-730); query and per-neighbor gather
                                run in libpnerf_hip.so
  query_dense(inputs)        -> the fused path's dense device tensors (no host sync)
  prune / grow_points / set_points / reset_querier : neural_points.py:341-467 (re-create the nn.Parameters,
                                which also invalidates the cached voxel grid: the cache is keyed on the storage)
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .point_query import lighting_fast_querier


class NeuralPoints(nn.Module):

    def __init__(self, num_channels, size, opt, device, checkpoint=None, feature_init_method='rand', reg_weight=0., feedforward=0):
        super().__init__()
        assert isinstance(size, int), 'size must be int'
        self.opt = opt
        self.grid_vox_sz = 0
        self.points_conf, self.points_dir, self.points_color, self.eulers, self.Rw2c = None, None, None, None, None
        self.xyz, self.points_embeding = None, None
        self.device = device
        if getattr(opt, "load_points", 0) == 1 and checkpoint is not None:
            saved = torch.load(checkpoint, map_location=device) if isinstance(checkpoint, str) else checkpoint
            get = lambda k: saved["neural_points." + k] if ("neural_points." + k) in saved else None
            self.xyz = nn.Parameter(get("xyz").to(device))
            self.xyz.requires_grad = opt.xyz_grad > 0
            for name, flag in (("points_embeding", "feat_grad"), ("points_conf", "conf_grad"), ("points_dir", "dir_grad"),
                               ("points_color", "color_grad")):
                t = get(name)
                if t is not None:
                    p = nn.Parameter(t.to(device))
                    p.requires_grad = getattr(opt, flag) > 0
                    setattr(self, name, p)
            rw = get("Rw2c")
            if rw is not None:
                self.Rw2c = nn.Parameter(rw.to(device)); self.Rw2c.requires_grad = False
            else:
                self.Rw2c = torch.eye(3, device=device, dtype=torch.float32)
        self.reg_weight = reg_weight
        self.opt.query_size = self.opt.kernel_size if self.opt.query_size[0] == 0 else self.opt.query_size
        if getattr(opt, "wcoord_query", -1) >= 0:
            raise NotImplementedError("only the world-coordinate torch-ext querier (wcoord_query=-1) is replaced; "
                                      "the pycuda queriers are out of scope (SURVEY.md 2 #4,#5)")
        self.lighting_fast_querier = lighting_fast_querier
        self.querier = self.lighting_fast_querier(device, self.opt)

    # ------------------------------------------------------------------ mutators (neural_points.py:341-467)
    def reset_querier(self):
        self.querier.clean_up()
        del self.querier
        self.querier = self.lighting_fast_querier(self.device, self.opt)

    def invalidate_grid(self):
        """Drop the cached voxel grids.  The cache recognises a changed cloud by (storage address, version counter, shape); writes
        that bypass the version counter -- ``xyz.data.copy_()`` / ``xyz.data.add_()`` from a caller's script, a raw-pointer kernel
        -- must be followed by this call, or the query keeps walking the old grid.  prune / grow_points / set_points call it."""
        from . import point_query
        point_query.clear_grid_cache()

    def _param(self, t, flag):
        p = nn.Parameter(t)
        p.requires_grad = getattr(self.opt, flag) > 0
        return p

    def prune(self, thresh):
        mask = self.points_conf[0, ..., 0] >= thresh
        self.xyz = self._param(self.xyz[mask, :], "xyz_grad")
        for name, flag in (("points_embeding", "feat_grad"), ("points_conf", "conf_grad"), ("points_dir", "dir_grad"),
                           ("points_color", "color_grad")):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, self._param(t[:, mask, :], flag))
        self.invalidate_grid()
        print("@@@@@@@@@  pruned {}/{}".format(torch.sum(mask == 0), mask.shape[0]))

    def grow_points(self, add_xyz, add_embedding, add_color, add_dir, add_conf, add_eulers=None, add_Rw2c=None):
        self.xyz = self._param(torch.cat([self.xyz, add_xyz], dim=0), "xyz_grad")
        for name, flag, add in (("points_embeding", "feat_grad", add_embedding), ("points_conf", "conf_grad", add_conf),
                                ("points_dir", "dir_grad", add_dir), ("points_color", "color_grad", add_color)):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, self._param(torch.cat([t, add[None, ...]], dim=1), flag))
        self.invalidate_grid()

    def set_points(self, points_xyz, points_embeding, points_color=None, points_dir=None, points_conf=None, parameter=False,
                   Rw2c=None, eulers=None):
        opt = self.opt
        if points_embeding.shape[-1] > opt.point_features_dim:
            points_embeding = points_embeding[..., :opt.point_features_dim]
        dc = getattr(opt, "default_conf", -1.0)
        if dc > 0.0 and dc <= 1.0 and points_conf is not None:
            points_conf = torch.ones_like(points_conf) * dc
        wrap = (lambda t, f: self._param(t, f)) if parameter else (lambda t, f: t)
        self.xyz = wrap(points_xyz, "xyz_grad")
        for t, mode, name, flag in ((points_conf, opt.point_conf_mode, "points_conf", "conf_grad"),
                                    (points_dir, opt.point_dir_mode, "points_dir", "dir_grad"),
                                    (points_color, opt.point_color_mode, "points_color", "color_grad")):
            if t is not None:
                t = wrap(t, flag)
                if "0" in list(mode):
                    points_embeding = torch.cat([t, points_embeding], dim=-1)
                if "1" in list(mode):
                    setattr(self, name, t)
        self.points_embeding = wrap(points_embeding, "feat_grad")
        self.invalidate_grid()
        if Rw2c is None:
            self.Rw2c = torch.eye(3, device=points_xyz.device, dtype=points_xyz.dtype)
        else:
            self.Rw2c = nn.Parameter(Rw2c)
            self.Rw2c.requires_grad = False

    # ------------------------------------------------------------------ hot path
    def w2pers(self, point_xyz, camrotc2w, campos):
        """neural_points.py:604-610."""
        point_xyz_shift = point_xyz[None, ...] - campos[:, None, :]
        xyz = torch.sum(camrotc2w[:, None, :, :] * point_xyz_shift[:, :, :, None], dim=-2)
        xper = xyz[:, :, 0] / xyz[:, :, 2]
        yper = xyz[:, :, 1] / xyz[:, :, 2]
        return torch.stack([xper, yper, xyz[:, :, 2]], dim=-1)

    def query_dense(self, inputs):
        """Fused-path query: dense [R,...] device tensors + the work list; no host synchronisation."""
        from . import ops
        near, far = float(ops.host_array(inputs["near"]).min()), float(ops.host_array(inputs["far"]).max())
        return self.querier.query_dense(self.xyz[None, ...], self.xyz.shape[0], near, far, inputs["raydir"], inputs["campos"])

    def get_point_indices(self, inputs, cam_rot_tensor, cam_pos_tensor, pixel_idx_tensor, near_plane, far_plane, h, w, intrinsic,
                          vox_query=False):
        """neural_points.py:555-577 (vox_query / NN<0 is not on the scripts' path and not supported)."""
        actual = torch.ones([1], device=self.xyz.device, dtype=torch.int32) * self.xyz.shape[0]
        sample_pidx, sample_loc, sample_loc_w, sample_ray_dirs, ray_mask, vsize, ranges = self.querier.query_points(
            pixel_idx_tensor, None, self.xyz[None, ...], actual, h, w, intrinsic, near_plane, far_plane, inputs["raydir"],
            cam_pos_tensor, cam_rot_tensor)
        return sample_pidx, sample_loc, ray_mask, None, sample_loc_w, sample_ray_dirs, vsize

    def forward(self, inputs):
        """The reference's 14-tuple (neural_points.py:699-730)."""
        pixel_idx, camrotc2w, campos = inputs["pixel_idx"].to(torch.int32), inputs["camrotc2w"], inputs["campos"]
        near_plane, far_plane = inputs["near"], inputs["far"]
        sample_pidx, sample_loc, ray_mask_tensor, _, sample_loc_w_tensor, sample_ray_dirs_tensor, vsize = self.get_point_indices(
            inputs, camrotc2w, campos, pixel_idx, torch.min(near_plane).cpu().numpy(), torch.max(far_plane).cpu().numpy(),
            None, None, None, vox_query=False)
        sample_pnt_mask = sample_pidx >= 0
        B, R, SR, K = sample_pidx.shape
        g = lambda t: None if t is None else ops.gather_rows(t.reshape(-1, t.shape[-1]), sample_pidx)
        sampled_xyz = g(self.xyz)
        xs = sampled_xyz - campos[:, None, None, None, :]
        xc = torch.sum(xs[..., None, :] * torch.transpose(camrotc2w, 1, 2)[:, None, None, None, ...], dim=-1)
        sampled_xyz_pers = torch.stack([xc[..., 0] / xc[..., 2], xc[..., 1] / xc[..., 2], xc[..., 2]], dim=-1)
        sampled_Rw2c = self.Rw2c
        return g(self.points_color), sampled_Rw2c, g(self.points_dir), g(self.points_conf), g(self.points_embeding), \
            sampled_xyz_pers, sampled_xyz, sample_pnt_mask, sample_loc, sample_loc_w_tensor, sample_ray_dirs_tensor, \
            ray_mask_tensor, vsize, self.grid_vox_sz
