"""A self-consistent optimisation problem for the convergence A/B of the weight-gradient arithmetic (tests/test_gpu_zz_convergence.py,
tools/gpu_convergence.py): a TEACHER (cloud attributes + MLP) renders ground-truth colours of a ring of views with the HIP inference
path; a perturbed STUDENT (same positions, embeddings / confidences / colours / directions moved, MLP pulled half-way to another
initialisation) is optimised against them with the loop body of the reference (models/mvs_points_volumetric_model.py:98-118: forward,
the lego script's losses, backward, both Adam steps; run/train_ft.py:829-937 is the loop around it): a fixed, seeded sequence of
(view, ray subset) batches, no sample jitter, so that two runs see identical inputs and differ only in arithmetic (and in the order of
the backward's atomics: that is the run-to-run spread the A/B is judged against)."""
import numpy as np
import torch

from oracle import pyref
from pointnerf_amd import config, scenes, ops
from pointnerf_amd import dist as pdist
from pointnerf_amd.neural_points import NeuralPoints
from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
from pointnerf_amd.optim import FusedAdam
from pointnerf_amd.point_aggregators import PointAggregator

TRAIN_THETAS = [0.0, 30.0, 60.0, 90.0, 120.0, 150.0, 180.0, 210.0, 240.0, 270.0, 300.0, 330.0]
HELD_THETAS = [15.0, 135.0, 255.0]
BLOCK = 72                       # 72 x 72 pixels around the image centre: the cloud's silhouette (radius ~28 px) and a rim of misses


def scene(n_points=20000, seed=0, K=8, SR=32, block=BLOCK, radius=0.10):
    opt = config.lego_opt(K=K, SR=SR, P=16, max_o=200000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    xyz = torch.from_numpy(scenes.chair_points(n_points, seed=seed, radius=radius))
    teacher = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n_points, 32, seed).items()}
    mlp_t = pyref.init_mlp_params(opt, seed=seed, bias_scale=0.1)
    x0 = 400 - block // 2
    views = lambda thetas: [pyref.to_torch_inputs(scenes.block_rays(theta_deg=t, x0=x0, y0=x0, size=block)) for t in thetas]
    # A freshly initialised network renders fog of one colour (sigma ~ 0.3 over 4 mm samples; rgb = sigmoid(~0)).  The teacher gets the heads of
    # a TRAINED model: density of the order of 100 (opaque within a few samples) and a colour head whose pre-activations are standardised over
    # the samples of view 0 (zero mean, unit deviation per channel: saturated neither way) -- calibrated with the CPU oracle, once
    g0 = torch.Generator().manual_seed(4321 + seed)
    mlp_t["alpha_branch.0.weight"] = mlp_t["alpha_branch.0.weight"] * 40.0
    mlp_t["alpha_branch.0.bias"] = mlp_t["alpha_branch.0.bias"] + 150.0
    mlp_t["color_branch.6.weight"] = 0.02 * torch.randn(mlp_t["color_branch.6.weight"].shape, generator=g0)
    mlp_t["color_branch.6.bias"] = torch.zeros_like(mlp_t["color_branch.6.bias"])
    cal = pyref.render(opt, dict(xyz=xyz, **teacher), mlp_t, views([TRAIN_THETAS[0]])[0], nthreads=8)
    valid = cal["query"]["sample_pidx"][0][..., 0] >= 0
    rgb = ((cal["decoded_features"][0][..., 1:][valid] + 0.001) / 1.002).clamp(1e-4, 1 - 1e-4)
    pre = torch.log(rgb) - torch.log1p(-rgb)
    mu, sd = pre.mean(0), pre.std(0).clamp(min=1e-6)
    mlp_t["color_branch.6.weight"] = mlp_t["color_branch.6.weight"] / sd[:, None]
    mlp_t["color_branch.6.bias"] = -mu / sd
    # student: attributes moved by a fixed perturbation, MLP half-way to another initialisation
    g = torch.Generator().manual_seed(1234 + seed)
    student = dict(points_embeding=teacher["points_embeding"] + 0.25 * torch.randn(teacher["points_embeding"].shape, generator=g),
                   points_conf=(teacher["points_conf"] + 0.2 * torch.randn(teacher["points_conf"].shape, generator=g)).clamp(0.05, 1.0),
                   points_color=(teacher["points_color"] + 0.2 * torch.randn(teacher["points_color"].shape, generator=g)).clamp(0.0, 1.0),
                   points_dir=torch.nn.functional.normalize(teacher["points_dir"] + 0.3 * torch.randn(teacher["points_dir"].shape, generator=g), dim=-1))
    other = pyref.init_mlp_params(opt, seed=seed + 77, bias_scale=0.1)
    mlp_s = {k: 0.5 * (mlp_t[k] + other[k]) for k in mlp_t}
    return opt, xyz, teacher, mlp_t, student, mlp_s, views(TRAIN_THETAS), views(HELD_THETAS)


def build_model(opt, xyz, attrs, mlp, dev):
    agg = PointAggregator(opt).to(dev)
    agg.load_state_dict(mlp)
    agg.flatten_()
    npnt = NeuralPoints(32, xyz.shape[0], opt, dev)
    a = {k: v.to(dev) for k, v in attrs.items()}
    npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"], parameter=True)
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt).to(dev)
    agg.flatten_()
    return model, agg, npnt


def render_view(model, view, dev):
    """colours of all rays of a view ([R, 3], background where nothing is hit): the inference forward"""
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in view.items()}
    with torch.no_grad():
        out = model(**d)
    col = d["bg_color"].expand(d["raydir"].shape[1], 3).clone()
    col[out["ray_mask"][0] > 0] = out["coarse_raycolor"][0]
    return col


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def run(dev, steps, planes, rays_per_step=1024, seed=0, log_every=100, sc=None):
    """optimise the student for `steps` steps with `planes` f16 planes per weight-gradient operand; returns the loss curve (means over
    `log_every` steps), the mean loss of the last 100 steps and the PSNR of the held-out views"""
    opt, xyz, teacher, mlp_t, student, mlp_s, train_views, held_views = sc if sc is not None else scene(seed=seed)
    old = ops.set_wgrad_planes(planes)
    try:
        t_model, _, _ = build_model(opt, xyz, teacher, mlp_t, dev)
        gts = [render_view(t_model, v, dev) for v in train_views]
        held_gt = [render_view(t_model, v, dev) for v in held_views]
        del t_model
        model, agg, npnt = build_model(opt, xyz, student, mlp_s, dev)
        model.fused_zero_one = True
        mlp_params = [p for p in agg.parameters() if p.requires_grad]
        pt_params = [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
        Adam = FusedAdam if torch.device(dev).type == "cuda" else torch.optim.Adam          # (the emulator dry run of this harness steps with torch's)
        o_mlp, o_pts = Adam(mlp_params, lr=opt.lr, betas=(0.9, 0.999)), Adam(pt_params, lr=opt.plr, betas=(0.9, 0.999))
        psnr0 = float(np.mean([psnr(render_view(model, v, dev), g) for v, g in zip(held_views, held_gt)]))
        mse_all = lambda: float(np.mean([float(((render_view(model, v, dev) - g) ** 2).mean()) for v, g in zip(train_views, gts)]))
        mse0 = mse_all()
        g = torch.Generator().manual_seed(99 + seed)
        nray = train_views[0]["raydir"].shape[1]
        curve, acc, last = [], [], []
        dviews = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in view.items()} for view in train_views]
        for step in range(steps):
            vi = step % len(train_views)
            sel = torch.randperm(nray, generator=g)[:rays_per_step].sort().values.to(dev)
            d = dict(dviews[vi])
            d["raydir"], d["pixel_idx"], d["gt_image"] = d["raydir"][:, sel], d["pixel_idx"][:, sel], gts[vi][sel][None]
            o_mlp.zero_grad(set_to_none=True); o_pts.zero_grad(set_to_none=True)
            out = model(**d)
            loss = pdist.hot_path_loss(opt, out, d["gt_image"])
            loss.backward()
            o_mlp.step(); o_pts.step()
            acc.append(loss.detach())
            if len(acc) == log_every or step == steps - 1:
                vals = torch.stack(acc).cpu()
                if not bool(torch.isfinite(vals).all()):
                    raise FloatingPointError("non-finite loss at step <= %d" % step)
                curve.append(float(vals.mean()))
                last = vals[-100:] if vals.numel() >= 100 else vals
                acc = []
        held = float(np.mean([psnr(render_view(model, v, dev), gt) for v, gt in zip(held_views, held_gt)]))
        mse1 = mse_all()                  # all rays of all training views, after the last step
        return dict(planes=planes, steps=steps, loss_curve=curve, final_loss=float(last.mean()), psnr_heldout=held, psnr_heldout_before=psnr0,
                    train_mse=mse1, train_mse_before=mse0, psnr_train=float(-10.0 * np.log10(mse1)))
    finally:
        ops.set_wgrad_planes(old)
