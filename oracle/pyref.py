"""oracle/pyref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (torch fp32, single precision end to end) restatement of everything on the hot path
that is *not* the native query op: grid hyper-parameters, ray sample generation, the
world->perspective projection, the per-neighbor gather, the aggregator (distance weights +
``viewmlp`` MLP), ray-dist, ``ray_march``, ``fill_invalid`` and the training loss.  Gradients
of the oracle come from torch.autograd over these functions.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference followed (under /root/reference), function by function:
  grid_hyperparameters     models/neural_points/point_query.py:27-71
  ray_samples              models/rendering/diff_ray_marching.py:349-392 (jitter=0)
  w2pers                   models/neural_points/neural_points.py:604-610, point_query.py:101-108
  gather_neighbors         models/neural_points/neural_points.py:699-730
  positional_encoding      models/helpers/networks.py:175-190
  aggregate                models/aggregators/point_aggregators.py:727-814 (+ linear :421-429,
                           viewmlp :488-644, raw2out_* :262-273, gradiant_clamp :722-724)
  ray_dist                 models/neural_points_volumetric_model.py:271-279
  ray_march                models/rendering/diff_ray_marching.py:508-554,
                           models/rendering/diff_render_func.py:36-62 (radiance / alpha / off)
  fill_invalid             models/neural_points_volumetric_model.py:87-123
  training_loss            models/base_rendering_model.py:543-551,630-641
  compute_losses           models/base_rendering_model.py:533-662 (colour / zero-one / sparse items)
  rank_ray_miss            models/mvs_points_volumetric_model.py:147-156
  probe_hole_mask          run/train_ft.py:489-500,532-540
  test_view_losses         run/train_ft.py:330-372 (per-view MSE / PSNR items of test())

Parity pin: the reference's own PointAggregator / ray_march / near_far_linear_ray_generation
import and run on CPU in the authoring container; tests/golden/make_golden.py stores their
outputs on seeded inputs and tests/test_oracle_golden.py checks this file against them.  The blocks of
modules that cannot be imported here (ray_dist, fill_invalid, the opt.prob == 1 outputs, probe_hole, test(),
construct_vox_points_closest) are exec'ed from the reference's SOURCE TEXT by the same script
(refblocks.npz, refshell.npz) and pinned by tests/test_oracle_golden.py and tests/test_reference_pins.py.
Only the lego-script configuration of the aggregator (agg_dist_pers=20,
linear kernel, agg_intrp_order=2, LeakyReLU, apply_pnt_mask=1, *_xyz_mode None) is restated.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import query as oq


# ----------------------------------------------------------------------------- grid / rays
def grid_hyperparameters(opt, xyz):
    """point_query.py:27-71.  xyz [N,3] f32 tensor -> dict of fp32/int32 numpy values."""
    vsize = np.asarray(opt.vsize, dtype=np.float64)                 # python floats -> f64 (point_query.py:69)
    vscale = np.asarray(opt.vscale, dtype=np.int32)
    scaled_vsize = (np.asarray(opt.vsize) * vscale).astype(np.float32)          # :37
    radius = np.asarray(opt.radius_limit_scale * max(opt.vsize[0], opt.vsize[1])).astype(np.float32)   # :35
    mn, mx = xyz.min(dim=0)[0], xyz.max(dim=0)[0]
    rmin = torch.as_tensor(opt.ranges[:3], dtype=torch.float32)
    rmax = torch.as_tensor(opt.ranges[3:], dtype=torch.float32)
    mn, mx = torch.maximum(mn, rmin), torch.minimum(mx, rmax)       # :62
    pad = torch.as_tensor(scaled_vsize * np.asarray(opt.kernel_size) / 2, dtype=torch.float32)   # :65 (f64 -> f32)
    mn, mx = mn - pad, mx + pad
    vdim = (mx - mn).numpy() / vsize                                # :69  f32 / f64 -> f64
    scaled_vdim = np.ceil(vdim / vscale).astype(np.int32)           # :70
    return dict(ranges=torch.cat([mn, mx]).numpy().astype(np.float32), scaled_vsize=scaled_vsize,
                scaled_vdim=scaled_vdim, radius=float(radius), vsize=np.asarray(opt.vsize))


def ray_samples(campos, raydir, D, near, far, jitter=0.0, uniforms=None):
    """diff_ray_marching.py:369-392.  campos [1,3], raydir [1,R,3] -> raypos [1,R,D,3], mid [D] (jitter = 0) or [1,R,D].
    jitter > 0: `uniforms` [1,R,D] plays the role of the reference's torch.rand (the HIP path draws them from a counter RNG,
    pnerf_debug_uniform); torch.cumsum on the CPU is the sequential fp32 running sum."""
    t = torch.linspace(0, 1, D + 1).view(1, -1)
    t = near * (1 - t) + far * t
    u = torch.zeros(1, 1, D) if uniforms is None else uniforms
    seg = (t[..., 1:] - t[..., :-1]) * (1 + jitter * (u - 0.5))
    end = torch.cumsum(seg, dim=2)
    end = near + torch.cat([torch.zeros(end.shape[0], end.shape[1], 1), end], dim=2)
    mid = (end[:, :, :-1] + end[:, :, 1:]) / 2
    raypos = campos[:, None, None, :] + raydir[:, :, None, :] * mid[:, :, :, None]
    return raypos, (mid.reshape(-1) if uniforms is None else mid)


def w2pers(p, camrot, campos):
    """p [...,3] world -> (x/z, y/z, z) in the camera frame.  camrot [3,3] is c2w, campos [3]."""
    c = (p - campos) @ camrot          # (R^T (p - o))_j = sum_i (p-o)_i R_ij
    return torch.stack([c[..., 0] / c[..., 2], c[..., 1] / c[..., 2], c[..., 2]], dim=-1)


def query(opt, xyz, inp, impl="oracle", nthreads=1, jitter=0.0, uniforms=None):
    """lighting_fast_querier.query_points (point_query.py:74-98) on CPU.
    Returns dict(sample_pidx [1,R2,SR,K] i32, sample_loc_w, sample_loc (pers), sample_ray_dirs, ray_mask [1,R] i8, hp)."""
    hp = grid_hyperparameters(opt, xyz)
    campos, raydir = inp["campos"], inp["raydir"]
    near, far = float(inp["near"].min()), float(inp["far"].max())
    raypos, _ = ray_samples(campos, raydir, opt.z_depth_dim, near, far, jitter=jitter, uniforms=uniforms)
    R = raydir.shape[1]
    fn = oq.oracle_query if impl == "oracle" else oq.ref_query
    pidx, loc_w, ray_mask, info = fn(raypos[0].numpy(), xyz.numpy(), opt.kernel_size, opt.query_size,
                                     opt.SR, opt.K, hp["scaled_vdim"], opt.max_o, opt.P, hp["radius"],
                                     hp["ranges"], hp["scaled_vsize"], nthreads=nthreads)
    pidx, loc_w, ray_mask = torch.from_numpy(pidx)[None], torch.from_numpy(loc_w)[None], torch.from_numpy(ray_mask)[None]
    dirs = raydir[0][ray_mask[0] > 0][None, :, None, :].expand(-1, -1, opt.SR, -1).contiguous()
    loc_p = w2pers(loc_w, inp["camrotc2w"][0], campos[0])
    return dict(sample_pidx=pidx, sample_loc_w=loc_w, sample_loc=loc_p, sample_ray_dirs=dirs,
                ray_mask=ray_mask, hp=hp, info=info, R=R)


# ----------------------------------------------------------------------------- aggregator
def positional_encoding(x, freqs, ori=False):
    """networks.py:175-190: per input dim d, per freq f: (sin, cos) interleaved; ori=True is
    [x | all sins | all coss]."""
    bands = 2.0 ** torch.arange(freqs, dtype=x.dtype)
    p = (x[..., None] * bands).reshape(x.shape[:-1] + (freqs * x.shape[-1],))
    if ori:
        return torch.cat([x, torch.sin(p), torch.cos(p)], dim=-1)
    return torch.stack([torch.sin(p), torch.cos(p)], dim=-1).reshape(p.shape[:-1] + (p.shape[-1] * 2,))


def mlp_param_shapes(opt):
    """state_dict key -> shape of the reference PointAggregator built from the lego flags
    (viewmlp_init, point_aggregators.py:276-348)."""
    Fd, H = opt.point_features_dim, opt.shading_feature_num
    dist_dim = 6 if opt.agg_dist_pers == 20 else 3
    in1 = Fd + 2 * opt.num_feat_freqs * Fd + 2 * abs(opt.dist_xyz_freq) * dist_dim      # 284
    in3 = H + 3 + 4                                                                       # 263
    inc = H + 2 * opt.num_viewdir_freqs * 3 + opt.view_ori * 3                            # 280
    Hc = H // 2
    s = {"block1.0": (H, in1), "block1.2": (H, H), "block3.0": (H, in3), "block3.2": (H, H),
         "alpha_branch.0": (1, H), "color_branch.0": (Hc, inc), "color_branch.2": (Hc, Hc),
         "color_branch.4": (Hc, Hc), "color_branch.6": (3, Hc)}
    out = {}
    for k, v in s.items():
        out[k + ".weight"] = v
        out[k + ".bias"] = (v[0],)
    return out


def init_mlp_params(opt, seed=0, bias_scale=0.0):
    """Xavier-uniform weights with the reference's gains (networks.py:163-172: leaky_relu gain for
    layers followed by LeakyReLU, gain 1 for the last layer of each Sequential), zero biases unless
    bias_scale>0 (tests use non-zero biases so that bias handling is exercised)."""
    g = torch.Generator().manual_seed(seed)
    gain_l = math.sqrt(2.0 / (1 + 0.01 ** 2))
    # in the reference every Linear that is followed by a LeakyReLU *inside its Sequential* gets the
    # leaky gain; the final Linear of each Sequential gets gain 1.  block1/block3 end with an activation
    # module, so their second Linear is also followed by LeakyReLU -> leaky gain; init_seq's trailing
    # init_weights(s[-1]) then hits the activation module (no-op).
    last = {"alpha_branch.0", "color_branch.6"}
    params = {}
    for k, shp in mlp_param_shapes(opt).items():
        if k.endswith(".weight"):
            gain = 1.0 if k[:-7] in last else gain_l
            bound = gain * math.sqrt(6.0 / (shp[0] + shp[1]))
            params[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        else:
            params[k] = (torch.rand(shp, generator=g) * 2 - 1) * bias_scale
    return params


def gather_neighbors(points, pidx, camrot, campos):
    """neural_points.py:699-717: -1 slots read point 0, validity lives in the mask only."""
    mask = pidx >= 0
    idx = pidx.clamp(min=0).long().view(-1)
    xyz = points["xyz"]
    xyz_pers = w2pers(xyz, camrot, campos)
    shp = pidx.shape
    g = lambda t: t.reshape(-1, t.shape[-1])[idx].view(shp + (t.shape[-1],))
    return dict(mask=mask, xyz=g(xyz), xyz_pers=g(xyz_pers), emb=g(points["points_embeding"][0]),
                color=g(points["points_color"][0]), dir=g(points["points_dir"][0]),
                conf=g(points["points_conf"][0]))


def aggregate(opt, mlp, nb, loc_p, loc_w, ray_dirs, Rw2c=None, kink=None):
    """PointAggregator.forward + viewmlp for the lego configuration.
    nb: dict from gather_neighbors with tensors [1,R,SR,K,*]; loc_* / ray_dirs [1,R,SR,3].
    Returns output [1,R,SR,4], ray_valid [1,R,SR] bool, weight [1,R,SR,K], conf_coefficient [1,R,SR,K].
    kink: optional dict that receives, for the tests' LeakyReLU-kink attribution, the smallest |pre-activation| of every
    neighbor row over the four viewmlp layers (`row_min_pre` [n rows], rows in mask order) and of every valid sample over the
    three colour layers (`sample_min_pre` [n valid samples])."""
    mask = nb["mask"]
    B, R, SR, K = mask.shape
    dt = loc_w.dtype
    Rw2c = torch.eye(3, dtype=dt) if Rw2c is None else Rw2c.to(dt)
    Rt = Rw2c.transpose(-1, -2)
    ray_valid = mask.any(dim=-1)
    out = torch.zeros(B, R, SR, 4, dtype=dt)
    xp, lp = nb["xyz_pers"], loc_p[..., None, :]
    dists = torch.cat([nb["xyz"] - loc_w[..., None, :],
                       torch.stack([xp[..., 0] * xp[..., 2] - lp[..., 0] * lp[..., 2],
                                    xp[..., 1] * xp[..., 2] - lp[..., 1] * lp[..., 2],
                                    xp[..., 2] - lp[..., 2]], dim=-1)], dim=-1)          # :773-781
    w = mask.to(dt) / torch.clamp(torch.linalg.norm(dists[..., :3], dim=-1), min=1e-6)     # linear :425-428
    w = w / torch.clamp(w.sum(dim=-1, keepdim=True), min=1e-8)                               # :801-802
    conf = nb["conf"][..., 0]
    conf_c = conf - (conf - conf.clamp(1e-4, 1.0)).detach()                                  # gradiant_clamp
    if ray_valid.sum() == 0:
        return out, ray_valid, w, conf_c
    wc = w * conf_c                                                                          # :811
    mf = mask.view(-1)
    vf = ray_valid.view(-1)
    # per-neighbor rows (compacted by the point mask, :523-538)
    d = dists.view(-1, 6)[mf]
    d = torch.cat([d[:, :3] @ Rt, d[:, 3:]], dim=-1)                                         # :526
    d = positional_encoding(d, opt.dist_xyz_freq)
    feat = nb["emb"].reshape(-1, nb["emb"].shape[-1])[mf]
    feat = torch.cat([feat, positional_encoding(feat, opt.num_feat_freqs), d], dim=-1)       # 284
    act = lambda x: F.leaky_relu(x, 0.01)
    tracked = {}

    def lin(x, k):
        y = F.linear(x, mlp[k + ".weight"], mlp[k + ".bias"])
        if kink is not None and (k.startswith("block") or k in ("color_branch.0", "color_branch.2", "color_branch.4")):
            grp = "row" if k.startswith("block") else "sample"
            m = y.detach().abs().min(dim=-1)[0]
            tracked[grp] = m if grp not in tracked else torch.minimum(tracked[grp], m)
        return y
    feat = act(lin(act(lin(feat, "block1.0")), "block1.2"))
    view = ray_dirs.reshape(-1, 3) @ Rt                                                      # :506
    view_pe = positional_encoding(view, opt.num_viewdir_freqs, ori=True)[:, 3:]              # 24
    view_k = view[:, None, :].expand(-1, K, -1).reshape(-1, 3)[mf]
    pdir = nb["dir"].reshape(-1, 3)[mf] @ Rt                                                 # :564-566
    feat = torch.cat([feat, nb["color"].reshape(-1, 3)[mf], pdir - view_k,
                      (pdir * view_k).sum(-1, keepdim=True)], dim=-1)                        # 263
    feat = act(lin(act(lin(feat, "block3.0")), "block3.2"))
    alpha = F.softplus(lin(feat, "alpha_branch.0") - 1)                                      # :262-265
    n_all = B * R * SR * K
    a_full = torch.zeros(n_all, 1, dtype=dt).index_put((mf.nonzero()[:, 0],), alpha)
    f_full = torch.zeros(n_all, feat.shape[-1], dtype=dt).index_put((mf.nonzero()[:, 0],), feat)
    wk = wc.reshape(-1, K, 1)
    sigma = (a_full.view(-1, K, 1) * wk).sum(dim=1)[vf]                                      # :608-614
    fs = (f_full.view(-1, K, feat.shape[-1]) * wk).sum(dim=1)[vf]                            # :622-628
    c = torch.cat([fs, view_pe[vf]], dim=-1)                                                 # 280
    c = act(lin(c, "color_branch.0")); c = act(lin(c, "color_branch.2")); c = act(lin(c, "color_branch.4"))
    rgb = torch.sigmoid(lin(c, "color_branch.6")) * (1 + 2 * 0.001) - 0.001                  # :269-273
    res = torch.cat([sigma, rgb], dim=-1)
    out = out.view(-1, 4).index_put((vf.nonzero()[:, 0],), res).view(B, R, SR, 4)
    if kink is not None:
        kink["row_min_pre"], kink["sample_min_pre"] = tracked["row"], tracked["sample"]
    return out, ray_valid, w, conf_c


# ----------------------------------------------------------------------------- renderer
def ray_dist(opt, loc_p, ray_valid):
    """neural_points_volumetric_model.py:271-279."""
    vs2 = float(opt.vsize[2])
    z = torch.cummax(loc_p[..., 2], dim=-1)[0]
    d = torch.cat([z[..., 1:] - z[..., :-1], torch.full(z.shape[:2] + (1,), vs2, dtype=z.dtype)], dim=-1)
    m = d < 1e-8
    if opt.raydist_mode_unit > 0:
        m = m | (d > 2 * vs2)
    m = m.to(d.dtype)
    d = d * (1.0 - m) + m * vs2
    return d * ray_valid.to(d.dtype)


def ray_march(rdist, ray_valid, feats, bg_color=None):
    """diff_ray_marching.py:508-554 with radiance_render / alpha_blend."""
    rgb = feats[..., 1:4]
    sigma = feats[..., 0] * ray_valid.to(feats.dtype)
    opacity = 1 - torch.exp(-sigma * rdist)
    acc = torch.cumprod(1.0 - opacity + 1e-10, dim=-1)
    bg_t = acc[:, :, [-1]]
    acc = torch.cat([torch.ones_like(acc[:, :, :1]), acc[:, :, :-1]], dim=-1)
    bw = (opacity * acc)[..., None]
    color = (rgb * bw).sum(dim=-2)
    if bg_color is not None:
        color = color + bg_color.to(color.dtype).view(bg_t.shape[0], 1, 3) * bg_t
    return color, rgb, opacity, acc, bw, bg_t


def render(opt, points, mlp, inp, impl="oracle", q=None, nthreads=1, Rw2c=None, kink=None):
    """NeuralPointsRayMarching.forward (neural_points_volumetric_model.py:252-329) on CPU.
    points: dict xyz [N,3], points_embeding [1,N,F], points_conf [1,N,1], points_dir/color [1,N,3]."""
    if q is None:
        with torch.no_grad():
            q = query(opt, points["xyz"].detach(), inp, impl=impl, nthreads=nthreads)
    camrot, campos = inp["camrotc2w"][0], inp["campos"][0]
    nb = gather_neighbors(points, q["sample_pidx"], camrot, campos)
    feats, ray_valid, w, conf_c = aggregate(opt, mlp, nb, q["sample_loc"], q["sample_loc_w"], q["sample_ray_dirs"], Rw2c=Rw2c, kink=kink)
    rd = ray_dist(opt, q["sample_loc"], ray_valid)
    color, _, opacity, acc, bw, bg_t = ray_march(rd, ray_valid, feats, inp["bg_color"])
    return dict(coarse_raycolor=color, coarse_point_opacity=opacity, coarse_is_background=bg_t,
                ray_mask=q["ray_mask"], weight=w, blend_weight=bw, conf_coefficient=conf_c,
                decoded_features=feats, ray_valid=ray_valid, ray_dist=rd, query=q,
                queried_shading=torch.logical_not(ray_valid.any(dim=-1, keepdim=True)).repeat(1, 1, 3).float())


def render_f64(opt, points, mlp, inp, q, Rw2c=None, kink=None):
    """The same renderer in float64 on the SAME query result q (indices and sample positions stay the fp32 ones): the yardstick
    against which the fp32 oracle's and the HIP path's own rounding (incl. LeakyReLU-kink flips) are measured by the tests."""
    d = lambda t: t.detach().double()
    pts = {k: (d(v).requires_grad_(v.requires_grad) if torch.is_floating_point(v) else v) for k, v in points.items()}
    m64 = {k: d(v).requires_grad_(v.requires_grad) for k, v in mlp.items()}
    i64 = {k: (d(v) if isinstance(v, torch.Tensor) and torch.is_floating_point(v) else v) for k, v in inp.items()}
    q64 = {k: (d(v) if isinstance(v, torch.Tensor) and torch.is_floating_point(v) else v) for k, v in q.items()}
    out = render(opt, pts, m64, i64, q=q64, Rw2c=Rw2c, kink=kink)
    return out, pts, m64


def fill_invalid(out, inp):
    """neural_points_volumetric_model.py:87-123: scatter the R'' hit rays back to all R rays."""
    mask = out["ray_mask"][0] > 0
    R = mask.numel()
    bgt = torch.ones(1, R, 1); bgt[0, mask] = out["coarse_is_background"][0]
    col = (torch.ones(1, R, 3) * inp["bg_color"][None]).clone(); col[0, mask] = out["coarse_raycolor"][0]
    op = torch.zeros(1, R, out["coarse_point_opacity"].shape[2]); op[0, mask] = out["coarse_point_opacity"][0]
    return dict(coarse_raycolor=col, coarse_is_background=bgt, coarse_mask=1 - bgt, coarse_point_opacity=op)


def training_loss(opt, out, inp, zero_epsilon=1e-3):
    """base_rendering_model.py:543-551 (ray_masked_coarse_raycolor, weight 1.0 + 1e-6) and :630-641
    (zero_one on conf_coefficient, weight opt.zero_one_loss_weights[0])."""
    mask = out["ray_mask"][0] > 0
    gt = inp["gt_image"][0][mask]
    pred = out["coarse_raycolor"][0]
    loss = F.mse_loss(pred, gt) if pred.shape[0] > 0 else torch.tensor(0.0)
    total = loss * 1.0 + 1e-6
    if "conf_coefficient" in opt.zero_one_loss_items and out["conf_coefficient"] is not None:
        v = out["conf_coefficient"].clamp(zero_epsilon, 1 - zero_epsilon)
        total = total + torch.mean(torch.log(v) + torch.log(1 - v)) * opt.zero_one_loss_weights[0]
    return total


def to_torch_inputs(d):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


def probe_outputs(out, points):
    """neural_points_volumetric_model.py:331-362 (opt.prob == 1) on the oracle's render() result.  Pinned bit for bit against the
    reference's own statements (exec'ed from the source text: tests/golden/refshell.npz, tests/test_reference_pins.py)."""
    q = out["query"]
    op = out["coarse_point_opacity"]
    mx, ind = torch.max(op, dim=-1, keepdim=True)
    ind = ind[..., None]
    loc_max = torch.gather(q["sample_loc_w"], 2, ind.expand(-1, -1, -1, 3)).squeeze(2)
    w = torch.gather(out["weight"] * out["conf_coefficient"], 2, ind.expand(-1, -1, -1, out["weight"].shape[-1])).squeeze(2)[..., None]
    pidx = torch.gather(q["sample_pidx"], 2, ind.expand(-1, -1, -1, q["sample_pidx"].shape[-1])).squeeze(2).clamp(min=0).long()
    g = lambda t: t.reshape(-1, t.shape[-1])[pidx.view(-1)].view(pidx.shape + (t.shape[-1],))
    xyz_max = g(points["xyz"])
    return dict(ray_max_shading_opacity=mx, ray_max_sample_loc_w=loc_max,
                ray_max_far_dist=torch.min(torch.norm(xyz_max - loc_max[..., None, :], dim=-1), dim=-1, keepdim=True)[0],
                shading_avg_color=torch.sum(g(points["points_color"][0]) * w, dim=-2),
                shading_avg_dir=torch.sum(g(points["points_dir"][0]) * w, dim=-2),
                shading_avg_conf=torch.sum(g(points["points_conf"][0]) * w, dim=-2),
                shading_avg_embedding=torch.sum(g(points["points_embeding"][0]) * w, dim=-2))


# ----------------------------------------------------------------------------- point initialisation
def vox_points_closest(xyz, vox_res, space_min=None, space_max=None):
    """models/mvs/mvs_utils.py:537-561 (construct_vox_points_closest) restated on CPU without torch_scatter, in the canonical
    order of the HIP path: voxels in torch.unique(dim=0) order, centroid = fp32 sum in point order / count, closest member by
    fp32 sqrt(dx^2 + dy^2 + dz^2), ties to the lowest index.  Points outside a caller-given box are dropped.
    Pinned against the reference function itself, exec'ed from its source text with pure-torch scatter_mean / scatter_min
    (tests/golden/refshell.npz, tests/test_reference_pins.py): voxel list, centroids and closest members identical."""
    xyz = xyz.float()
    if space_min is None:
        xyz_min, xyz_max = torch.min(xyz, dim=-2)[0], torch.max(xyz, dim=-2)[0]
        space_edge = torch.max(xyz_max - xyz_min) * 1.05
        space_min = (xyz_max + xyz_min) / 2 - space_edge / 2
        vox = (space_edge / vox_res).expand(3)
    else:
        space_min, space_max = torch.as_tensor(space_min).float(), torch.as_tensor(space_max).float()
        vox = (space_max - space_min) / vox_res
    cell = torch.floor((xyz - space_min[None]) / vox[None]).to(torch.int64)
    inside = ((cell >= 0) & (cell < vox_res)).all(dim=-1)
    key = (cell[:, 0] * vox_res + cell[:, 1]) * vox_res + cell[:, 2]
    order = torch.argsort(torch.where(inside, key, torch.full_like(key, -1)), stable=True)
    order = order[int((~inside).sum()):]
    ks = key[order]
    bounds = torch.nonzero(torch.cat([torch.tensor([True]), ks[1:] != ks[:-1]])).reshape(-1).tolist() + [len(order)]
    cen, gidx, midx = [], [], []
    x32 = xyz.numpy()
    import numpy as _np
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        mem = order[b0:b1].tolist()
        s = _np.zeros(3, dtype=_np.float32)
        for p in mem:
            s = (s + x32[p]).astype(_np.float32)
        c = (s / _np.float32(len(mem))).astype(_np.float32)
        best, arg = _np.float32(3.402823466e38), -1
        for p in mem:
            d = (x32[p] - c).astype(_np.float32)
            r = _np.sqrt(_np.float32(_np.float32(_np.float32(d[0] * d[0]) + _np.float32(d[1] * d[1])) + _np.float32(d[2] * d[2])))
            if r < best:
                best, arg = r, p
        cen.append(c); midx.append(arg)
        k = int(ks[b0])
        gidx.append([k // (vox_res * vox_res), (k // vox_res) % vox_res, k % vox_res])
    return (torch.from_numpy(_np.stack(cen)), torch.tensor(gidx, dtype=torch.int32), torch.tensor(midx, dtype=torch.int64),
            int((~inside).sum()))



# ----------------------------------------------------------------------------- model shell (loss items, ray-miss ranking, probe)
def compute_losses(out, gt_image, color_items, color_weights, zero_one_items=(), zero_one_weights=(), zero_epsilon=1e-3,
                   sparse_loss_weight=0.0):
    """models/base_rendering_model.py:533-662 on the scattered [1,R,*] outputs, written with the reference's own ops
    (masked_select + MSELoss mean).  Returns (loss_total, {name: loss})."""
    l2 = torch.nn.MSELoss()
    total, parts = 0, {}
    for i, name in enumerate(color_items):
        if name.startswith("ray_masked"):
            key = name[len("ray_masked") + 1:]
            m = (out["ray_mask"] > 0)[..., None].expand(-1, -1, 3)
            po, pg = torch.masked_select(out[key], m).reshape(1, -1, 3), torch.masked_select(gt_image, m).reshape(1, -1, 3)
            loss = l2(po, pg) if po.shape[1] > 0 else torch.tensor(0.0)
        elif name.startswith("ray_miss"):
            key = name[len("ray_miss") + 1:]
            m = (out["ray_mask"] == 0)[..., None].expand(-1, -1, 3)
            po, pg = torch.masked_select(out[key], m).reshape(1, -1, 3), torch.masked_select(gt_image, m).reshape(1, -1, 3)
            loss = l2(po, pg) * pg.shape[1] if po.shape[1] > 0 else torch.tensor(0.0)
        else:
            loss = l2(out[name], gt_image)
        total = total + (loss * color_weights[i] + 1e-6)
        parts[name] = loss
    for i, name in enumerate(zero_one_items):
        if name not in out:
            continue
        val = torch.clamp(out[name], zero_epsilon, 1 - zero_epsilon)
        loss = torch.mean(torch.log(val) + torch.log(1 - val))
        total = total + loss * zero_one_weights[i]
        parts[name] = loss
    if sparse_loss_weight > 0:
        loss = torch.sum(out["weight"] * torch.abs(1 - torch.exp(-2 * out["conf_coefficient"]))) / (torch.sum(out["weight"]) + 1e-6)
        total = total + loss * sparse_loss_weight
        parts["sparse"] = loss
    return total, parts


def rank_ray_miss(new_id, newloss, inds, losses):
    """models/mvs_points_volumetric_model.py:147-156 with python lists (stable descending sort is NOT implied by the
    reference's torch.sort; tests use distinct losses)."""
    inds, losses = list(inds), list(losses)
    if new_id in inds:
        j = inds.index(new_id)
        losses[j] = max(newloss, losses[j])
    else:
        inds[-1], losses[-1] = new_id, newloss
    order = sorted(range(len(losses)), key=lambda j: -losses[j])
    return [losses[j] for j in order], [inds[j] for j in order]


def probe_hole_mask(ray_mask, opacity, far_dist, raycolor, gt, bg, edge, opacity_thresh, far_thresh=-1.0):
    """run/train_ft.py:489-500 + bloat_inds :532-540 as index loops over numpy [H,W,*] maps.  Pinned against the reference's
    probe_hole exec'ed from its source text on stand-in model / dataset objects (tests/test_reference_pins.py)."""
    H, W = edge.shape
    miss = (ray_mask < 1) & (np.linalg.norm(gt - bg, axis=-1) > 0.002) & edge
    near = np.zeros((H, W), np.float32)
    for r, c in zip(*np.nonzero(miss)):
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                near[min(max(r + dr, 0), H - 1), min(max(c + dc, 0), W - 1)] = 1
    if far_thresh > 0:
        near = near + ((ray_mask > 0) & (far_dist > far_thresh) & (np.linalg.norm(gt - raycolor, axis=-1) < 0.1))
    return (ray_mask > 0) & (near > 0) & (opacity > opacity_thresh)


def test_view_losses(canvas, gt_rays, pixel_idx, ray_mask, height, width):
    """run/train_ft.py:330-372 for one view with numpy: canvas [H,W,3] rendered colours, gt_rays [P,3] and pixel_idx [P,2]
    (px, py) in row-major pixel order, ray_mask [P] bool.  Returns {item: mse, item_psnr: psnr}.  Pinned against the reference's
    test() exec'ed from its source text on stand-in model / dataset / visualizer objects (tests/test_reference_pins.py)."""
    edge = np.zeros((height, width), bool)
    edge[pixel_idx[:, 1], pixel_idx[:, 0]] = True
    gt = np.zeros((height * width, 3), np.float32)
    gt[edge.reshape(-1)] = gt_rays
    full = float(np.mean((canvas.reshape(-1, 3).astype(np.float64) - gt) ** 2))
    pred = canvas.reshape(-1, 3)[edge.reshape(-1)][ray_mask]
    masked = float(np.mean((pred.astype(np.float64) - gt_rays[ray_mask]) ** 2))
    ps = lambda x: -10.0 * math.log(x) / math.log(10.0)
    return dict(coarse_raycolor=full, coarse_raycolor_psnr=ps(full), ray_masked_coarse_raycolor=masked, ray_masked_coarse_raycolor_psnr=ps(masked))


# ---- point initialisation from given 2-D maps --------------------------------------------------------------------------
def project_to_view(cam_xyz, c2w, w2c, intrinsic, HD, WD, depth_occ, tolerate=0.1):
    """models/mvs/mvs_utils.py homo_warp_nongrid :299-315 (depth_occ == 0) / homo_warp_nongrid_occ :333-369 in numpy fp32 WITHOUT the
    compaction: -> (pixel [N,2], mask [N]).  cam_xyz [N,3] in the frame of the camera c2w belongs to; w2c None = that camera itself.
    Pinned through query_embedding against the reference's functions exec'ed from source (tests/golden/refembed.npz)."""
    f32 = np.float32
    p = cam_xyz.astype(f32)
    if w2c is not None:
        h = np.concatenate([p, np.ones((len(p), 1), f32)], 1)
        p = ((h @ c2w.T.astype(f32)).astype(f32) @ w2c.T.astype(f32)).astype(f32)[:, :3]
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (p / p[:, 2:3]).astype(f32)
    pix = (q @ intrinsic.T.astype(f32)).astype(f32)[:, :2]
    lim = np.array([WD - 1, HD - 1], f32)
    if not depth_occ:
        mask = np.all((pix >= 0) & (pix <= lim), axis=1)
        return pix, mask
    mask = np.all((pix >= 0) & (np.ceil(pix) <= lim), axis=1)
    hard = np.ceil(pix[mask]).astype(np.int64)
    cell = hard[:, 0] * HD + hard[:, 1]                                    # :355 (x * HD + y)
    z = p[mask, 2]
    zmin = np.full(int(cell.max()) + 1 if len(cell) else 0, np.inf, f32)
    np.minimum.at(zmin, cell, z)
    keep = z <= zmin[cell] + f32(tolerate)                                 # :361
    out = np.zeros(len(p), bool)
    out[np.nonzero(mask)[0][keep]] = True
    return pix, out


def grid_sample_bilinear(fmap, pix, HD, WD):
    """extract_from_2d_grid (models/mvs/mvs_utils.py:411-415): F.grid_sample(bilinear, zeros, align_corners=True) of fmap [C,H,W] at the
    pixel positions scaled to [-1, 1] by (WD-1, HD-1) (:313-314) -> [N,C]"""
    f32 = np.float32
    C, H, W = fmap.shape
    gx = pix[:, 0] / f32((WD - 1.0) / 2.0) - f32(1)
    gy = pix[:, 1] / f32((HD - 1.0) / 2.0) - f32(1)
    x = (gx + f32(1)) * f32((W - 1) / 2.0)
    y = (gy + f32(1)) * f32((H - 1) / 2.0)
    x0, y0 = np.floor(x), np.floor(y)
    w, n = x - x0, y - y0
    e, s = f32(1) - w, f32(1) - n
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def tex(ix, iy):
        ok = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
        v = fmap[:, np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1)].T
        return np.where(ok[:, None], v, f32(0))
    return (tex(x0, y0) * (s * e)[:, None] + tex(x0 + 1, y0) * (s * w)[:, None] + tex(x0, y0 + 1) * (n * e)[:, None]
            + tex(x0 + 1, y0 + 1) * (n * w)[:, None]).astype(f32)


def extract_2d(img_feats, view_ids, layer_ids, intrinsics, c2ws, w2cs, cam_xyz, HD, WD, cam_vid=0, depth_occ=0):
    """MvsPointsModel.extract_2d (models/mvs/mvs_points_model.py:198-218), batch of one, numpy: img_feats = list over pyramid layers of
    [V,C,H,W]; intrinsics [V,3,3], c2ws / w2cs [V,4,4], cam_xyz [N,3] -> (out_feats [N, sum], colors [N, 3 per view] or None)"""
    feats, colors = [], []
    for vid in view_ids:
        pix, mask = project_to_view(cam_xyz, c2ws[cam_vid], None if vid == cam_vid else w2cs[vid], intrinsics[vid], HD, WD, depth_occ)
        pix = np.where(mask[:, None], pix, np.float32(0))                                   # NaN / inf of rejected points
        for lid in layer_ids:
            v = grid_sample_bilinear(img_feats[lid][vid], pix, HD, WD) * mask[:, None]
            (colors if lid == 0 else feats).append(v.astype(np.float32))
    return np.concatenate(feats, -1), (np.concatenate(colors, -1) if colors else None)


def point_dirs(cam_xyz, view_ids, c2ws, w2cs, cam_vid, ref_vid=0, pointdir_w=False):
    """the "dir" block of query_embedding (models/mvs/mvs_points_model.py:239-251) -> [N, 3 per view]"""
    f32 = np.float32
    pos_w = c2ws[view_ids][:, :, 3]                                                            # [V,4]
    pos_c = (pos_w @ w2cs[cam_vid].T).astype(f32)[:, :3]
    d = cam_xyz[:, None, :] - pos_c[None]
    d = d / (np.sqrt((d * d).sum(-1, keepdims=True)).astype(f32) + f32(1e-6))
    d = (d.reshape(-1, 3) @ c2ws[cam_vid][:3, :3].T).astype(f32)
    if not pointdir_w:
        d = (d @ c2ws[ref_vid][:3, :3].T).astype(f32)
    return d.reshape(len(cam_xyz), -1)


def query_embedding(feat_strs, HDWD, cam_xyz, photometric_confidence, img_feats, c2ws, w2cs, intrinsics, cam_vid, pointdir_w=False,
                    depth_occ=0, ref_vid=0):
    """MvsPointsModel.query_embedding (models/mvs/mvs_points_model.py:225-259) with shading_feature_mlp_layer0 == 0, batch of one."""
    emb, colors, dirs, conf = [], None, None, None
    for s in feat_strs:
        if s.startswith("imgfeat"):
            _, v, l = s.split("_")
            f, colors = extract_2d(img_feats, [int(a) for a in v], [int(a) for a in l], intrinsics, c2ws, w2cs, cam_xyz, HDWD[0], HDWD[1],
                                   cam_vid=cam_vid, depth_occ=depth_occ)
            emb.append(f)
        elif s.startswith("dir"):
            dirs = point_dirs(cam_xyz, np.array([int(a) for a in s.split("_")[1]]), c2ws, w2cs, cam_vid, ref_vid, pointdir_w)
        elif s.startswith("point_conf"):
            conf = np.ones_like(emb[0][:, :1]) if photometric_confidence is None else photometric_confidence
    return np.concatenate(emb, -1), colors, dirs, conf
