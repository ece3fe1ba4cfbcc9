"""Drop-in for the reference's ``models/aggregators/point_aggregators.py`` class ``PointAggregator`` -- the
lego-script configuration (``which_agg_model=viewmlp``, ``agg_distance_kernel=linear``,
``agg_dist_pers=20``, ``agg_intrp_order=2``, ``act_type=LeakyReLU``, ``apply_pnt_mask=1``, ``*_xyz_mode=None``,
``num_feat_freqs=3``, ``dist_xyz_freq=5``, ``num_viewdir_freqs=4``, 256-wide feature MLPs).

Sub-module names and shapes equal the reference's (``viewmlp_init``, point_aggregators.py:276-348), so
``state_dict()`` keys are ``block1.0.weight`` ... ``color_branch.6.bias`` and reference checkpoints load with
``load_state_dict``.  All parameters are views into ONE flat fp32 vector in the layout libpnerf_hip.so expects
(``pnerf_mlp_layout``), so the kernels read the live weights and optimizers update them in place.
Initialisation restates ``init_seq`` (models/helpers/networks.py:163-172).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .fused import MLPState


def _check_supported(opt):
    want = dict(which_agg_model="viewmlp", agg_distance_kernel="linear", agg_dist_pers=20, agg_intrp_order=2,
                act_type="LeakyReLU", apply_pnt_mask=1, agg_feat_xyz_mode="None", agg_alpha_xyz_mode="None",
                agg_color_xyz_mode="None", num_feat_freqs=3, dist_xyz_freq=5, num_viewdir_freqs=4, point_features_dim=32,
                shading_feature_num=256, shading_feature_mlp_layer1=2, shading_feature_mlp_layer2=0,
                shading_feature_mlp_layer3=2, shading_alpha_mlp_layer=1, shading_color_mlp_layer=4, agg_weight_norm=1,
                act_super=1, dist_xyz_deno=0, view_ori=0, shading_color_channel_num=3)
    bad = {k: getattr(opt, k, None) for k, v in want.items() if getattr(opt, k, v) != v}
    if bad:
        raise NotImplementedError("pointnerf_amd.PointAggregator implements the lego-script aggregator only "
                                  "(SURVEY.md 8); unsupported option values: %r" % bad)
    aw = getattr(opt, "agg_axis_weight", None)
    if aw is not None and not (float(aw[0]) == 1.0 and float(aw[2]) == 1.0):
        raise NotImplementedError("agg_axis_weight other than 1 1 1 is not on the scripts' path")


class PointAggregator(nn.Module):

    def __init__(self, opt):
        super().__init__()
        _check_supported(opt)
        self.opt = opt
        act = lambda: nn.LeakyReLU(inplace=True)
        H, Hc = 256, 128
        self.block1 = nn.Sequential(nn.Linear(284, H), act(), nn.Linear(H, H), act())
        self.block3 = nn.Sequential(nn.Linear(263, H), act(), nn.Linear(H, H), act())
        self.alpha_branch = nn.Sequential(nn.Linear(H, 1))
        self.color_branch = nn.Sequential(nn.Linear(280, Hc), act(), nn.Linear(Hc, Hc), act(), nn.Linear(Hc, Hc), act(),
                                          nn.Linear(Hc, 3))
        for seq in (self.block1, self.block3, self.alpha_branch, self.color_branch):
            self._init_seq(seq)
        self._flat = None
        self._state = None

    @staticmethod
    def _init_seq(s):
        """networks.py:163-172: xavier_uniform with the leaky_relu gain for a Linear followed by LeakyReLU,
        gain 1 otherwise; biases zero."""
        mods = list(s)
        for i, m in enumerate(mods):
            if not isinstance(m, nn.Linear):
                continue
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            gain = nn.init.calculate_gain('leaky_relu', nxt.negative_slope) if isinstance(nxt, nn.LeakyReLU) else 1.0
            nn.init.xavier_uniform_(m.weight, gain=gain)
            nn.init.constant_(m.bias, 0.0)

    # ---- flat storage -------------------------------------------------------------------------
    def _named_layout(self):
        lay, total = ops.mlp_layout()
        return lay, total

    def flatten_(self):
        """Move every parameter into one flat device vector (idempotent; call after .to(device)/load_state_dict)."""
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        ops._need_cuda(next(iter(params.values())), "PointAggregator parameters (call .to('cuda') before use)")
        lay, total = self._named_layout()
        base = self._flat
        already = base is not None and base.device == dev and all(
            params[k].data_ptr() == base.data_ptr() + 4 * o for k, (o, shp) in lay.items())
        if not already:
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            for k, (o, shp) in lay.items():
                n = int(np.prod(shp))
                flat[o:o + n].copy_(params[k].data.reshape(-1))
                params[k].data = flat[o:o + n].view(shp)
            self._flat = flat
            self._state = MLPState(flat)
            self.check_range()
        return self._flat

    def check_range(self):
        """The two-plane f16 form of the GEMM operands (csrc/f16x3.h) holds |x| <= 65504; the fp32 GEMMs it replaces had the whole
        fp32 range.  Weights are checked whenever they are (re)homed into the flat vector -- after ``.to(device)`` / ``load_state_dict``,
        i.e. when a checkpoint arrives -- and a violation raises instead of rendering NaN.  (Activations of this network are O(1..100);
        gradients are clamped by the kernels.)  One host read; not on the step path."""
        if self._flat is None:
            return
        amax = float(self._flat.detach().abs().max())
        if not (amax <= 65504.0):                       # also catches NaN / inf
            raise ValueError("PointAggregator: a weight of magnitude %g (or a non-finite one) exceeds the range of the two-plane f16 "
                             "arithmetic of libpnerf_hip.so (|w| <= 65504)" % amax)

    def mlp_state(self):
        self.flatten_()
        return self._state

    def ordered_params(self):
        """(parameters in pnerf_mlp_layout order, [(offset, numel, shape)])."""
        lay, _ = self._named_layout()
        params = dict(self.named_parameters())
        return [params[k] for k in lay], [(o, int(np.prod(shp)), shp) for k, (o, shp) in lay.items()]

    def flat_grad_to_params(self, gflat):
        """Attach slices of a flat gradient vector as .grad of the individual parameters (accumulating)."""
        lay, _ = self._named_layout()
        params = dict(self.named_parameters())
        for k, (o, shp) in lay.items():
            g = gflat[o:o + int(np.prod(shp))].view(shp)
            p = params[k]
            if p.requires_grad:
                p.grad = g.clone() if p.grad is None else p.grad + g

    def forward(self, sampled_color, sampled_Rw2c, sampled_dir, sampled_conf, sampled_embedding, sampled_xyz_pers, sampled_xyz,
                sample_pnt_mask, sample_loc, sample_loc_w, sample_ray_dirs, vsize, grid_vox_sz):
        """Reference signature (point_aggregators.py:727-814).  Returns (output [B,R,SR,4], ray_valid [B,R,SR] bool,
        weight | None, conf_coefficient | None) with the reference's rule for the last two (:812-813)."""
        from .fused import Aggregate
        from .neural_points_volumetric_model import gradient_clamp
        opt = self.opt
        B, R, SR, K = sample_pnt_mask.shape
        assert B == 1, "batch size 1 (every reference script)"
        in_shape = sample_loc_w.shape
        ray_valid = torch.any(sample_pnt_mask, dim=-1)
        dev = sampled_embedding.device
        if R == 0 or not bool(ray_valid.any()):
            return torch.zeros(in_shape[:-1] + (4,), device=dev, dtype=torch.float32), ray_valid, None, None
        if sampled_Rw2c is not None and sampled_Rw2c.dim() != 2:
            raise NotImplementedError("per-point Rw2c (normview) is not on the scripts' path")
        if sampled_color is None or sampled_dir is None or sampled_conf is None:
            raise NotImplementedError("point_color/dir/conf_mode must be '1' (lego script)")
        st = self.mlp_state()
        mlp_params, layout = self.ordered_params()
        rw = None if sampled_Rw2c is None else sampled_Rw2c.detach().cpu().numpy()
        cam = ops.make_camera([0, 0, 0], np.eye(3), opt.vsize[2], 1, bg=None, rw2c=rw)     # camera unused: perspective coords supplied
        n_slots = R * SR * K
        slot = torch.arange(n_slots, dtype=torch.int32, device=dev).view(R, SR, K)
        pidx = torch.where(sample_pnt_mask[0], slot, torch.full_like(slot, -1)).contiguous()
        c = lambda t, w: t.detach().reshape(-1, w).contiguous().float()
        env = dict(cam=cam, xyz_slots=c(sampled_xyz, 3), xyz_pers=c(sampled_xyz_pers, 3), loc_w=c(sample_loc_w, 3),
                   loc_pers=c(sample_loc, 3), raydir=sample_ray_dirs[0, :, 0, :].detach().contiguous().float(), pidx=pidx,
                   nn=sample_pnt_mask[0].sum(-1).to(torch.int32).contiguous(), R=R, SR=SR, K=K, flat=st.flat,
                   packed=st.packed_image(), train=torch.is_grad_enabled(), layout=layout)
        decoded, weight = Aggregate.apply(env, sampled_embedding, sampled_conf, sampled_dir, sampled_color, *mlp_params)
        conf_coefficient = gradient_clamp(sampled_conf[..., 0], lo=0.0001, hi=1)
        weight = weight.view(B, R, SR, K)
        if (opt.sparse_loss_weight <= 0) and ("conf_coefficient" not in opt.zero_one_loss_items) and getattr(opt, "prob", 0) == 0:
            weight, conf_coefficient = None, None
        return decoded.view(in_shape[:-1] + (4,)), ray_valid, weight, conf_coefficient
