"""Drop-in for ``NeuralPointsRayMarching`` of the reference's ``models/neural_points_volumetric_model.py``
(:220-364) -- THE hot nn.Module (query -> aggregate -> ray_dist -> ray_march -> output dict) -- and for
``fill_invalid`` (:87-123).

``forward(**input)`` takes the reference's kwargs (``campos, raydir, gt_image, bg_color, camrotc2w, pixel_idx, near,
far, focal, h, w, intrinsic, **kargs``) and returns the reference's dict with the reference's shapes
(``coarse_raycolor [1,R'',3]``, ``coarse_point_opacity [1,R'',SR]``, ``coarse_is_background [1,R'',1]``,
``ray_mask [1,R] int8``, ``queried_shading``, ``weight``, ``blend_weight``, ``conf_coefficient``).  Internally
everything runs dense over the R submitted rays inside libpnerf_hip.so; the R'' view is produced by one
boolean row-gather at the very end (the only host sync besides reading the valid-sample count).
``opt.prob==1`` adds the probe outputs of :331-362 (per-ray argmax-opacity sample, its location, nearest-neighbor
distance and weighted average colour/dir/conf/embedding) used by ``probe_hole`` (run/train_ft.py:417-530): they touch
one sample per ray, so they are small gathers on top of the dense render (``pnerf_gather_rows`` for the point rows).
"""
import os

import torch
import torch.nn as nn

from . import dist as pdist
from . import ops
from .fused import FusedRender


def gradient_clamp(sampled_conf, lo=0.0001, hi=1.0):
    """point_aggregators.py:722-724: clamp forward, identity backward."""
    diff = sampled_conf - torch.clamp(sampled_conf, min=lo, max=hi)
    return sampled_conf - diff.detach()


# PNERF_SPECULATE=0: every training step waits for its counters before it is enqueued (rounds 1-3)
SPECULATE = os.environ.get("PNERF_SPECULATE", "1") != "0"
# PNERF_ZERO_ONE_IN_RENDER=0 (dev A/B): the fused zero-one regulariser as its own forward / backward pass (round 3) instead of inside the render node
ZERO_ONE_IN_RENDER = os.environ.get("PNERF_ZERO_ONE_IN_RENDER", "1") != "0"


class NeuralPointsRayMarching(nn.Module):

    def __init__(self, tonemap_func=None, render_func=None, blend_func=None, aggregator=None, is_compute_depth=False,
                 neural_points=None, opt=None, num_pos_freqs=0, num_viewdir_freqs=0, **kwargs):
        super().__init__()
        self.aggregator = aggregator
        self.neural_points = neural_points
        self.opt = opt
        self.num_pos_freqs, self.num_viewdir_freqs = num_pos_freqs, num_viewdir_freqs
        self.render_func, self.blend_func, self.tone_map = render_func, blend_func, tonemap_func
        self.return_depth = is_compute_depth
        self.return_color = True
        if is_compute_depth:
            raise NotImplementedError("compute_depth references an undefined ray_ts in the reference "
                                      "(neural_points_volumetric_model.py:318-322); unsupported (SURVEY.md A.11 iii)")
        if opt is not None:
            if getattr(opt, "which_render_func", "radiance") != "radiance" or getattr(opt, "which_blend_func", "alpha") != "alpha" \
                    or getattr(opt, "which_tonemap_func", "off") != "off":
                raise NotImplementedError("only radiance / alpha / off (every script's setting) is implemented")
        self.last_stats = None
        self._pool_rays = 0
        self._pinned_words = None

    def _host_words(self, n):
        """pinned host memory for the step's asynchronous read-back of its counters"""
        if self._pinned_words is None or self._pinned_words.numel() != n:
            self._pinned_words = torch.empty(n, dtype=torch.int64).pin_memory()
        return self._pinned_words

    def render_dense(self, campos, raydir, camrotc2w, near, far, bg_color=None, train=None):
        """The fused step on all R rays.  Returns (ray_color [R,3], opacity, bg_trans, blend_w, decoded, weight, zo_sum, dense); zo_sum =
        the zero-one regulariser's numerator over the hit rays' conf_coefficient when ``_zero_one_in_render()`` (else a constant 0)."""
        opt, npnt, agg = self.opt, self.neural_points, self.aggregator
        train = torch.is_grad_enabled() if train is None else train
        if train and getattr(opt, "xyz_grad", 0) > 0:
            raise NotImplementedError("xyz_grad > 0 (optimising point positions) is not on any reference script's path and the "
                                      "fused backward produces no d/d xyz")
        R = raydir.reshape(-1, 3).shape[0]
        if train and self._pool_rays < R:              # worst case (every ray hits): ~16 live [R,SR,K] fp32 tensors around the loss
            ops.reserve_pool(16 * R * int(opt.SR) * int(opt.K) * 4, raydir.device)
            self._pool_rays = R
        dense = npnt.query_dense(dict(campos=campos, raydir=raydir, near=near, far=far))
        # data-parallel callers that exchange touched rows only (dist.plan_sparse_exchange) prepare the row list here, so that its two
        # counts travel with the one host read below instead of synchronising a second time after the backward
        plan = getattr(self, "plan_sparse", None)
        plan = plan(dense) if (plan is not None and train) else None
        words = dense["counters"].to(torch.int64) if plan is None else torch.cat([dense["counters"].to(torch.int64), plan[1]])
        st = agg.mlp_state()
        rw = ops.host_array(npnt.Rw2c) if isinstance(npnt.Rw2c, torch.Tensor) else None
        cam = ops.make_camera(ops.host_array(campos).reshape(-1)[:3], ops.host_array(camrotc2w).reshape(-1)[:9],
                              opt.vsize[2], opt.raydist_mode_unit,
                              bg=None if bg_color is None else ops.host_array(bg_color).reshape(-1)[:3], rw2c=rw)
        mlp_params, layout = agg.ordered_params()
        env = dict(cam=cam, xyz=npnt.xyz.detach().reshape(-1, 3).contiguous(), raydir=raydir.detach().reshape(-1, 3).contiguous().float(),
                   dense=dense, R=R, SR=int(opt.SR), K=int(opt.K), n_valid=0, flat=st.flat, packed=st.packed_image(),
                   train=bool(train), layout=layout, want_grad_event=bool(train) and raydir.is_cuda and pdist.active())
        if train and self._zero_one_in_render():
            env["zero_one_eps"] = float(getattr(opt, "zero_epsilon", 1e-3))
        leaves = (npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color) + tuple(mlp_params)
        # The step's one host read (number of valid samples: sizes the activation arena; number of hit rays: shapes of the outputs).
        # Round 4: a TRAINING step whose arena already exists is enqueued BEFORE that read with the arena's capacity as the bound -- every
        # kernel takes the actual counts from the device (`counters`, the class partition's tile counts), the host number is only "how much
        # scratch is there" -- so the device starts the aggregator right behind the query instead of idling through the host's wake-up and
        # its ~20 launches (0.2-0.3 ms per step in the kernel trace).  The counts arrive in pinned memory meanwhile; should they exceed the
        # capacity (a batch larger than every one before it), the speculative result is dropped and the step runs again after the arena grew.
        cap = ops.ARENA.capacity_samples(env["K"], raydir.device) if (train and raydir.is_cuda and SPECULATE) else 0
        out = None
        if cap > 0:
            host = self._host_words(words.numel())
            host.copy_(words, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
            env["n_valid"] = cap
            out = FusedRender.apply(env, *leaves)
            ready.synchronize()
            got = host.clone()
        else:
            got = words.cpu()                                 # the one sync
        n_valid = int(got[0])
        dropped = out is not None and n_valid > cap
        if dropped:                                           # (rare: the arena has to grow; nothing of the dropped result is used)
            ops.ARENA.give(env.pop("_saved", None))
            out, env = None, dict(env)
        if plan is not None:
            self.sparse_plan = (plan[0], int(got[8]), int(got[9]))
        self.last_stats = dict(n_valid_samples=n_valid, rays_hit=int(got[1]), n_selected=int(got[2]), n_neighbor_rows=int(got[3]), rays=R,
                               enqueued_before_host_read=out is not None, speculative_result_dropped=dropped)
        if out is None:
            env["n_valid"] = n_valid
            out = FusedRender.apply(env, *leaves)
        return out + (dense,)

    def _zero_one_in_render(self):
        """the zero-one regulariser on conf_coefficient as part of the render node (``fused_zero_one`` callers whose only consumer of
        conf_coefficient is that loss): its numerator is an output of the node, its conf gradient rides on the node's own conf atomics"""
        opt = self.opt
        return ZERO_ONE_IN_RENDER and bool(getattr(self, "fused_zero_one", False)) and opt.sparse_loss_weight <= 0 and getattr(opt, "prob", 0) == 0 \
            and "conf_coefficient" in getattr(opt, "zero_one_loss_items", ())

    def forward(self, campos, raydir, gt_image=None, bg_color=None, camrotc2w=None, pixel_idx=None, near=None, far=None,
                focal=None, h=None, w=None, intrinsic=None, **kargs):
        opt = self.opt
        if "bg_ray" in kargs:
            bg_color = None
        ray_color, opacity, bg_trans, blend_w, decoded, weight, zo_sum, dense = self.render_dense(campos, raydir, camrotc2w, near, far, bg_color)
        zo_in_render = torch.is_grad_enabled() and self._zero_one_in_render()
        hit = dense["ray_hit"] > 0
        SR, K = int(opt.SR), int(opt.K)
        # Indices of the hit rays WITHOUT a host round trip: their number is already on the host (the counters the arena was
        # sized from), so a stable sort of the 0/1 flags yields them in ascending order.  Boolean-mask indexing would
        # synchronise once per tensor (nonzero), and the device would idle between forward, loss and backward while the
        # host catches up; with this the whole step is enqueued behind one synchronisation.
        n_hit = self.last_stats["rays_hit"]
        # ``fused_color_loss`` (ours, like ``fused_zero_one`` below; set by callers whose loss goes through dist.hot_path_loss, and by the model
        # shell of this package when every colour-loss item with a non-zero weight is a ray_masked / ray_miss one -- the lego script's
        # setting: MvsPointsVolumetricModel.create_network_models): a TRAINING step whose only consumer of the rendered colours is the colour loss gets the dense ray colours and the
        # hit flags under "_dense_color" (ops.ColorLossRays: one pass forward, one backward, d colour written for every ray) and the compacted
        # [1, R'', ...] outputs are not formed -- no argsort, no index_selects, no scatter-back in the backward (~25 launches per step)
        if getattr(self, "fused_color_loss", False) and torch.is_grad_enabled() and getattr(opt, "prob", 0) == 0 \
                and opt.sparse_loss_weight <= 0 and getattr(self, "fused_zero_one", False):
            output = {"_dense_color": (ray_color, dense["ray_hit"], n_hit), "ray_mask": hit.to(torch.int8)[None],
                      "_dense_aux": (opacity.detach(), bg_trans.detach())}       # (for fill_invalid's full-size visuals: references, no work)
            if "conf_coefficient" in opt.zero_one_loss_items:
                if zo_in_render:
                    output["_zero_one_sum"] = (zo_sum, n_hit * SR * K)
                else:
                    output["_zero_one"] = (self.neural_points.points_conf, dense["sample_pidx"], dense["ray_hit"], n_hit * SR * K)
            return output
        idx = torch.argsort(dense["ray_hit"], descending=True, stable=True)[:n_hit]
        take = lambda t: t.index_select(0, idx)
        output = {"_hit_index": idx}
        # queried_shading = not any(ray_valid) per ray (:322 of the reference): the R'' rays ARE the rays with a valid sample (ray_mask comes from
        # the same neighbor table, query_worldcoords.cu:425-429), so it is identically zero -- no pass over the [R'', SR] counts
        output["queried_shading"] = torch.zeros(1, n_hit, 3, dtype=torch.float32, device=ray_color.device)
        output["coarse_raycolor"] = take(ray_color)[None]
        output["coarse_point_opacity"] = take(opacity)[None]
        output["coarse_is_background"] = take(bg_trans)[None, :, None]
        output["ray_mask"] = hit.to(torch.int8)[None]
        want_w = (opt.sparse_loss_weight > 0) or ("conf_coefficient" in opt.zero_one_loss_items) or getattr(opt, "prob", 0) != 0
        # ``fused_zero_one`` (ours; set by callers whose loss goes through dist.hot_path_loss / the model shell of this package): when the
        # only consumer of conf_coefficient is the zero-one regulariser, hand out what that loss needs -- (points_conf, the hit rays'
        # neighbor table) under "_zero_one" -- instead of materialising weight / conf_coefficient [1, R'', SR, K] (ops.ZeroOneConf)
        only_zero_one = getattr(self, "fused_zero_one", False) and opt.sparse_loss_weight <= 0 and getattr(opt, "prob", 0) == 0
        if want_w and only_zero_one:
            # (points_conf, the DENSE neighbor table, the rays' hit flags, number of conf_coefficient elements): the loss runs over the hit rays
            # of the dense table in place (ops.ZeroOneConfRays) -- no [R'', SR, K] copy of it
            if zo_in_render:        # (training: the numerator came out of the render node; evaluation under no_grad: the stand-alone pass)
                output["_zero_one_sum"] = (zo_sum, n_hit * SR * K)
            else:
                output["_zero_one"] = (self.neural_points.points_conf, dense["sample_pidx"], dense["ray_hit"], n_hit * SR * K)
        elif want_w:
            output["weight"] = take(weight)[None].detach()
            output["blend_weight"] = take(blend_w)[None, ..., None].detach()
            conf = self.neural_points.points_conf
            pidx_hit = take(dense["sample_pidx"])
            output["conf_coefficient"] = gradient_clamp(ops.gather_rows(conf.reshape(-1, 1), pidx_hit)[..., 0])[None]
        if getattr(opt, "prob", 0) == 1 and output["coarse_point_opacity"].shape[1] > 0:
            self._probe_outputs(output, dense, hit)
        return output

    def _probe_outputs(self, output, dense, hit):
        """neural_points_volumetric_model.py:331-362, same keys and shapes."""
        npnt = self.neural_points
        with torch.no_grad():
            op = output["coarse_point_opacity"][0]                                   # [R'',SR]
            mx, ind = torch.max(op, dim=-1, keepdim=True)                            # [R'',1]
            output["ray_max_shading_opacity"] = mx[None]
            loc_w = dense["sample_loc"][hit]                                         # [R'',SR,3]
            sel = lambda t: torch.gather(t, 1, ind.view(-1, 1, *([1] * (t.dim() - 2))).expand(-1, 1, *t.shape[2:])).squeeze(1)
            loc_max = sel(loc_w)                                                     # [R'',3]
            output["ray_max_sample_loc_w"] = loc_max[None]
            w = sel(output["weight"][0] * output["conf_coefficient"][0].detach())    # [R'',K]
            pidx = sel(dense["sample_pidx"][hit])                                    # [R'',K]  (-1 slots read point 0, as :708)
            g = lambda t: ops.gather_rows(t.detach().reshape(-1, t.shape[-1]), pidx)  # [R'',K,C]
            xyz_max = g(npnt.xyz)
            output["ray_max_far_dist"] = torch.min(torch.norm(xyz_max - loc_max[:, None, :], dim=-1), dim=-1, keepdim=True)[0][None]
            wk = w[..., None]
            output["shading_avg_color"] = torch.sum(g(npnt.points_color) * wk, dim=-2)[None]
            output["shading_avg_dir"] = torch.sum(g(npnt.points_dir) * wk, dim=-2)[None]
            output["shading_avg_conf"] = torch.sum(g(npnt.points_conf) * wk, dim=-2)[None]
            output["shading_avg_embedding"] = torch.sum(g(npnt.points_embeding) * wk, dim=-2)[None]


PROBE_KEYS = ("ray_max_sample_loc_w", "ray_max_shading_opacity", "shading_avg_color", "shading_avg_dir", "shading_avg_conf",
              "shading_avg_embedding", "ray_max_far_dist")


def fill_invalid(output, bg_color, tonemap_func=None, bg_ray=None, prob=0):
    """neural_points_volumetric_model.py:87-123: scatter the hit rays' results back to all R submitted rays (missed rays get
    the background colour / transmittance 1 / opacity 0; with ``prob == 1`` the probe outputs are zero-filled, :121-122
    ``unmask``).  ``bg_ray`` [B,R,3] replaces the constant background like :104-106."""
    ray_mask = output["ray_mask"]
    B, OR = ray_mask.shape
    if "_dense_color" in output:
        # the fused-colour-loss form of a training step (NeuralPointsRayMarching.forward): the renderer's results are DENSE over the R rays
        # already, so "filling" is a select per ray -- no scatter, no index tensor.  The filled tensors are detached: the colour loss takes
        # its gradient through ops.ColorLossRays on the dense colours (MvsPointsVolumetricModel.compute_losses), these are the visuals.
        ray_color, ray_hit, _ = output["_dense_color"]
        opacity, bg_trans = output["_dense_aux"]
        dev = ray_color.device
        hitb = (ray_hit > 0)
        bgt = torch.where(hitb, bg_trans, torch.ones((), dtype=torch.float32, device=dev))[None, :, None]
        if bg_ray is not None:
            col = bgt * bg_ray.to(dev) + torch.where(hitb[:, None], ray_color.detach(), torch.zeros((), dtype=torch.float32, device=dev))[None]
        else:
            bg = torch.ones([OR, 3], dtype=torch.float32, device=dev) * bg_color.to(dev).reshape(-1, 3)
            if tonemap_func is not None:
                bg = tonemap_func(bg)
            col = torch.where(hitb[:, None], ray_color.detach(), bg)[None]
        op = torch.where(hitb[:, None], opacity, torch.zeros((), dtype=torch.float32, device=dev))[None]
        qs = torch.where(hitb[:, None], torch.zeros((), dtype=torch.float32, device=dev), torch.ones([OR, 3], dtype=torch.float32, device=dev))[None]
        out = dict(output)
        out.update(coarse_is_background=bgt, coarse_mask=1 - bgt, coarse_raycolor=col, coarse_point_opacity=op, queried_shading=qs)
        return out
    sel = output["_hit_index"] if "_hit_index" in output else ray_mask[0] > 0      # index tensor: no nonzero() synchronisation
    dev = output["coarse_raycolor"].device
    bgt = torch.ones([B, OR, 1], dtype=torch.float32, device=dev)
    bgt[0, sel] = output["coarse_is_background"][0]
    if bg_ray is not None:
        col = bgt * bg_ray.to(dev)
        col[0, sel] = col[0, sel] + output["coarse_raycolor"][0]
    else:
        col = torch.ones([B, OR, 3], dtype=torch.float32, device=dev) * bg_color[None, ...].to(dev)
        if tonemap_func is not None:
            col = tonemap_func(col)
        col[0, sel] = output["coarse_raycolor"][0]
    op = torch.zeros([B, OR, output["coarse_point_opacity"].shape[2]], dtype=torch.float32, device=dev)
    op[0, sel] = output["coarse_point_opacity"][0]
    qs = torch.ones([B, OR, 3], dtype=torch.float32, device=dev)
    qs[0, sel] = output["queried_shading"][0]
    out = dict(output)
    out.update(coarse_is_background=bgt, coarse_mask=1 - bgt, coarse_raycolor=col, coarse_point_opacity=op, queried_shading=qs)
    if prob == 1 and "ray_max_shading_opacity" in output:
        for k in PROBE_KEYS:
            if output.get(k) is not None:
                t = torch.zeros([B, OR, *output[k].shape[2:]], dtype=output[k].dtype, device=dev)
                t[0, sel] = output[k][0]
                out[k] = t
    return out
