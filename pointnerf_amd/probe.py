"""Probe-and-grow step of the training loop: ``probe_hole`` of the reference (run/train_ft.py:417-530)
renders training views with ``opt.prob = 1`` and proposes new neural points where rays that HIT the cloud sit next to rays
that MISSED it although the ground truth is not background ("holes"), or -- with ``far_thresh > 0`` -- where the most
opaque shading location of a well-rendered ray is far from every neural point.

Same selection rule, same outputs (add_xyz, add_embedding, add_color, add_dir, add_conf); different in HOW:
  * a view is rendered in chunks as large as the dense query buffers allow (default 160 000 rays) instead of <= 32^2 = 1024
    rays per ``model.test()`` call, results are scattered into per-view maps by ``pixel_idx`` on the device;
  * the 3x3 dilation of the missed-ray mask is one ``max_pool2d`` (``bloat_inds`` below is kept for the callers that use it
    directly and for the test that shows the two agree);
  * nothing is written to disk (the reference saves a debug image per view through its Visualizer; pass ``on_view`` to get
    the rendered map instead).
One quirk of the reference is reproduced on purpose because it changes the result: ``add_conf`` is multiplied by
``opt.prob_mul`` once per processed view AFTER concatenation (train_ft.py:505), so candidates found in earlier views are
scaled again by every later view.
"""
import random

import numpy as np
import torch
import torch.nn.functional as F

PROBE_MAP_KEYS = ("coarse_raycolor", "ray_mask", "ray_max_sample_loc_w", "ray_max_far_dist", "ray_max_shading_opacity",
                  "shading_avg_color", "shading_avg_dir", "shading_avg_conf", "shading_avg_embedding")


def bloat_inds(inds, shift, height, width):
    """[N,2] (row, col) -> [N*(2 shift+1)^2, 2]: every index with all its (2 shift+1)^2 neighbours, clamped to the image
    (run/train_ft.py:532-540)."""
    r = torch.arange(-shift, shift + 1, dtype=torch.long, device=inds.device)
    sx, sy = torch.meshgrid(r, r, indexing="ij")
    out = (inds[:, None, :] + torch.stack([sx, sy], dim=-1).reshape(1, -1, 2)).reshape(-1, 2)
    out[..., 0] = torch.clamp(out[..., 0], min=0, max=height - 1)
    out[..., 1] = torch.clamp(out[..., 1], min=0, max=width - 1)
    return out


def hole_mask(prob_maps, gt_image, bg, edge_mask, opacity_thresh, far_thresh=-1.0):
    """The per-view candidate mask [H,W] bool (run/train_ft.py:489-500).

    prob_maps: dict of [H,W,C] maps (``ray_mask`` [H,W,1] 0/1, ``ray_max_shading_opacity``, ``ray_max_far_dist``,
    ``coarse_raycolor``); gt_image [H,W,3] (zero outside ``edge_mask``); bg [1,3]; edge_mask [H,W] bool = pixels the view
    provides rays for."""
    H, W = edge_mask.shape
    hit = prob_maps["ray_mask"][..., 0] > 0
    miss = torch.logical_not(hit) & (torch.norm(gt_image - bg, dim=-1) > 0.002) & edge_mask
    near_miss = F.max_pool2d(miss[None, None].float(), kernel_size=3, stride=1, padding=1)[0, 0]      # == bloat_inds(miss, 1)
    if far_thresh > 0:
        far = hit & (prob_maps["ray_max_far_dist"][..., 0] > far_thresh) & \
            (torch.norm(gt_image - prob_maps["coarse_raycolor"], dim=-1) < 0.1)
        near_miss = near_miss + far.float()
    return hit & (near_miss > 0) & (prob_maps["ray_max_shading_opacity"][..., 0] > opacity_thresh)


def select_probe_frames(model, n_views, opt, rng=random):
    """Which training views get probed (run/train_ft.py:441-456): the views with the largest missed-ray loss when the model
    keeps that ranking, else a random subset of ``n_views // prob_num_step`` views."""
    max_num = n_views // opt.prob_num_step
    if opt.prob_top == 1 and opt.prob_mode <= 0 and getattr(model, "top_ray_miss_ids", None) is not None:
        mask = model.top_ray_miss_loss[:-1] > 0.0
        return [int(i) for i in model.top_ray_miss_ids[:-1][mask][:max_num].tolist()], True
    ids = list(range(n_views))[:max_num]
    rng.shuffle(ids)
    return ids[:max_num], False


@torch.no_grad()
def render_probe_maps(model, view, height, width, chunk=160000):
    """Render one view with ``opt.prob == 1`` and scatter every probe output into an [H,W,C] map (train_ft.py:466-487)."""
    dev = model.device
    raydir = view["raydir"].to(dev)
    pixel_idx = view["pixel_idx"].to(dev)
    pixel_idx = pixel_idx.reshape(pixel_idx.shape[0], -1, pixel_idx.shape[-1])
    total = pixel_idx.shape[1]
    maps = {}
    for k in range(0, total, chunk):
        data = dict(view)
        data["raydir"] = raydir[:, k:k + chunk, :]
        data["pixel_idx"] = pixel_idx[:, k:k + chunk, :]
        if "gt_image" in view:
            data["gt_image"] = view["gt_image"][:, k:k + chunk, :]
        model.set_input(data)
        out = model.test()
        pid = data["pixel_idx"].to(torch.long)
        for key in PROBE_MAP_KEYS:
            if "ray_max_shading_opacity" not in out and key != "coarse_raycolor":
                break
            v = out[key][..., None] if key == "ray_mask" else out[key]
            if v is None:
                maps[key] = None
                continue
            if key not in maps:
                maps[key] = torch.zeros((height, width, v.shape[-1]), device=dev, dtype=v.dtype)
            maps[key][pid[0, :, 1], pid[0, :, 0], :] = v[0]
    return maps, pixel_idx


@torch.no_grad()
def probe_hole(model, views, opt, height, width, test_steps=0, opacity_thresh=0.7, frame_ids=None, chunk=160000, on_view=None):
    """Returns (add_xyz [M,3], add_embedding [M,F], add_color [M,3], add_dir [M,3], add_conf [M,1]) on the device.

    ``views``: a sequence (or anything with ``__getitem__``/``__len__``) of the datasets' per-view dicts in ``no_crop`` form:
    raydir [1,P,3], pixel_idx [1,h,w,2] or [1,P,2] (px, py), gt_image [1,P,3], bg_color [1,3], campos, camrotc2w, near, far,
    id."""
    dev = model.device
    Fdim = opt.point_features_dim
    add = dict(xyz=torch.zeros([0, 3], device=dev), conf=torch.zeros([0, 1], device=dev), color=torch.zeros([0, 3], device=dev),
               dir=torch.zeros([0, 3], device=dev), emb=torch.zeros([0, Fdim], device=dev))
    saved = dict(query_size=model.opt.query_size, prob=getattr(model.opt, "prob", 0), no_loss=getattr(model.opt, "no_loss", 0))
    if getattr(opt, "prob_kernel_size", None) is not None:
        tier = int(np.sum(np.asarray(opt.prob_tiers) < test_steps))
        model.opt.query_size = np.asarray(opt.prob_kernel_size[tier * 3:tier * 3 + 3])
    model.opt.prob = 1
    model.opt.no_loss = 1
    if frame_ids is None:
        frame_ids, _ = select_probe_frames(model, len(views), opt)
    try:
        for i in frame_ids:
            view = views[i]
            bg = view["bg_color"].to(dev).reshape(1, 3)
            maps, pixel_idx = render_probe_maps(model, view, height, width, chunk)
            if "ray_max_shading_opacity" not in maps:         # no ray of this view hit anything
                continue
            edge = torch.zeros([height, width], dtype=torch.bool, device=dev)
            pl = pixel_idx[0].to(torch.long)
            edge[pl[:, 1], pl[:, 0]] = True
            gt = torch.zeros((height * width, 3), dtype=torch.float32, device=dev)
            gt[edge.reshape(-1)] = view["gt_image"].to(dev).reshape(-1, 3)
            m = hole_mask(maps, gt.reshape(height, width, 3), bg, edge, opacity_thresh, getattr(opt, "far_thresh", -1.0))
            add["xyz"] = torch.cat([add["xyz"], maps["ray_max_sample_loc_w"][m]], dim=0)
            add["conf"] = torch.cat([add["conf"], maps["shading_avg_conf"][m]], dim=0) * opt.prob_mul
            add["color"] = torch.cat([add["color"], maps["shading_avg_color"][m]], dim=0)
            add["dir"] = torch.cat([add["dir"], maps["shading_avg_dir"][m]], dim=0)
            add["emb"] = torch.cat([add["emb"], maps["shading_avg_embedding"][m]], dim=0)
            if on_view is not None:
                on_view(i, maps, m)
    finally:
        model.opt.query_size, model.opt.prob, model.opt.no_loss = saved["query_size"], saved["prob"], saved["no_loss"]
    if opt.prob_mode == 0 and opt.prob_num_step > 1 and hasattr(model, "num_probe"):
        model.reset_ray_miss_ranking()
    return add["xyz"], add["emb"], add["color"], add["dir"], add["conf"]


def prune_and_grow_step(model, views, opt, total_steps, height, width, real_start=-1):
    """The two maintenance steps of the training loop around the hot path (run/train_ft.py:834-842 prune, :844-880 probe
    and grow), with the optimizer / scheduler rebuilds they need.  Returns the number of points added (the reference saves
    a checkpoint and exits after a grow so the next launch restarts from it; the caller decides here)."""
    added = 0
    if opt.prune_iter > 0 and real_start != total_steps and total_steps % opt.prune_iter == 0 \
            and 0 < total_steps < (opt.maximum_step - 1) and total_steps <= opt.prune_max_iter:
        with torch.no_grad():
            model.clean_optimizer()
            model.clean_scheduler()
            model.prune_points(opt.prune_thresh)
            model.setup_optimizer(opt)
            model.init_scheduler(total_steps, opt)
    if opt.prob_freq > 0 and real_start != total_steps and total_steps % opt.prob_freq == 0 and 0 < total_steps < (opt.maximum_step - 1):
        pk = getattr(opt, "prob_kernel_size", None)
        tier = int(np.sum(np.asarray(opt.prob_tiers) < total_steps)) if pk is not None else 0
        worst = float(model.top_ray_miss_loss[0]) if getattr(model, "top_ray_miss_loss", None) is not None else 0.0
        if (worst > 1e-5 or opt.prob_mode != 0 or opt.far_thresh > 0) and (pk is None or tier < (len(pk) // 3)):
            is_train = model.opt.is_train
            model.opt.is_train = 0
            model.eval()
            try:
                xyz, emb, color, dirs, conf = probe_hole(model, views, opt, height, width, test_steps=total_steps,
                                                         opacity_thresh=opt.prob_thresh)
            finally:
                model.opt.is_train = is_train
                model.train()
            if len(xyz) > 0:
                model.clean_optimizer_scheduler()
                model.grow_points(xyz, emb, color, dirs, conf)
                model.setup_optimizer(opt)
                model.init_scheduler(total_steps, opt)
                added = len(xyz)
    return added
