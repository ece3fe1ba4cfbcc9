"""GPU tests of the model shell (``create_model`` -> MvsPointsVolumetricModel, what run/train_ft.py drives) and of the probe step
on the device, against the CPU oracle:
  * ``optimize_parameters`` x 3 (forward through the HIP kernels, the script's three colour items + the zero-one item, backward,
    two one-pass Adam instances) vs the oracle's render + the reference's loss formulas + torch.optim.Adam;
  * ``probe_hole`` on a view whose ray block straddles the silhouette of the cloud vs the oracle's probe outputs pushed through
    the index-loop restatement of run/train_ft.py:489-505."""
import numpy as np
import pytest
import torch

from cases import CASES
from oracle import pyref
from pointnerf_amd import config, probe, scenes
from pointnerf_amd.mvs_points_volumetric_model import create_model

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _scene(name, tmp_path, **kw):
    ov, n, size, seed = CASES[name]
    opt = config.lego_train_opt(**ov, gpu_ids=[0], checkpoints_dir=str(tmp_path), num_point=n, default_conf=-1.0, **kw)   # keep the seeded confidences
    xyz = torch.from_numpy(scenes.chair_points(n, seed=seed, radius=0.06))
    attrs = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, opt.point_features_dim, seed).items()}
    mlp = pyref.init_mlp_params(opt, seed=seed, bias_scale=0.1)
    m = create_model(opt)
    m.aggregator.load_state_dict(mlp)
    m.aggregator.flatten_()
    a = {k: v.to(DEV) for k, v in attrs.items()}
    m.set_points(xyz.to(DEV), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"])
    return opt, m, xyz, attrs, mlp, size, seed


def test_optimize_parameters_matches_the_oracle_loop(tmp_path):
    STEPS = 3
    opt, m, xyz, attrs, mlp, size, seed = _scene("small_k8", tmp_path, prob_freq=100, prob_num_step=1, ray_jitter=0.0)
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=30.0 + 40 * seed, x0=400 - size // 2, y0=400 - size // 2, size=size))
    m.setup(opt, train_len=10)
    m.train()
    # ---- oracle: the reference's loop body on the CPU
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    oa = {k: v.clone().requires_grad_(True) for k, v in attrs.items()}
    o_mlp = torch.optim.Adam(list(om.values()), lr=opt.lr, betas=(0.9, 0.999))
    o_pts = torch.optim.Adam(list(oa.values()), lr=opt.plr, betas=(0.9, 0.999))
    s_mlp = torch.optim.lr_scheduler.LambdaLR(o_mlp, lambda it: pow(opt.lr_decay_exp, it / opt.lr_decay_iters))
    s_pts = torch.optim.lr_scheduler.LambdaLR(o_pts, lambda it: pow(opt.lr_decay_exp, it / opt.lr_decay_iters))
    ref_tot, ref_parts = [], []
    for _ in range(STEPS):
        o_mlp.zero_grad(); o_pts.zero_grad()
        out = pyref.render(opt, dict(xyz=xyz, **oa), om, inp, nthreads=8)
        full = dict(out); full.update(pyref.fill_invalid(out, inp))
        total, parts = pyref.compute_losses(full, inp["gt_image"], opt.color_loss_items, opt.color_loss_weights, opt.zero_one_loss_items,
                                            opt.zero_one_loss_weights, opt.zero_epsilon)
        total.backward()
        o_mlp.step(); o_pts.step(); s_mlp.step(); s_pts.step()
        ref_tot.append(float(total)); ref_parts.append({k: float(v) for k, v in parts.items()})
    # ---- device: the shell's own loop body (run/train_ft.py:756-765)
    R = inp["raydir"].shape[1]
    for step in range(STEPS):
        data = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
        data["id"] = torch.tensor([3])
        m.set_input(data)
        m.optimize_parameters(total_steps=step)
        m.update_learning_rate(opt=opt, total_steps=step)
        losses = {k: float(v) for k, v in m.get_current_losses().items()}
        assert abs(losses["total"] - ref_tot[step]) <= 1e-4 * max(1.0, abs(ref_tot[step])), (step, losses, ref_tot)
        for k, v in ref_parts[step].items():
            assert abs(losses[k] - v) <= 1e-4 * max(1.0, abs(v)), (step, k, losses[k], v)
        assert m.output["coarse_raycolor"].shape == (1, R, 3) and m.output["coarse_point_opacity"].shape == (1, R, opt.SR)
        assert m.output["ray_mask"].shape == (1, R) and m.coarse_raycolor is m.output["coarse_raycolor"]
        # the lego script's items (ray_masked 1.0, ray_miss 0.0, full image 0.0) take the fused colour loss over the renderer's dense ray colours:
        # no compaction of the hit rays in a training step, the filled image is a per-ray select (round 5)
        assert m.net_ray_marching.fused_color_loss and "_dense_color" in m._raw and "coarse_raycolor" not in m._raw and "_hit_index" not in m._raw
    assert abs(float(m.top_ray_miss_loss[0]) - max(p["ray_miss_coarse_raycolor"] for p in ref_parts)) <= 1e-4 * max(1.0, ref_parts[0]["ray_miss_coarse_raycolor"])
    assert abs(m.optimizer.param_groups[0]["lr"] - o_mlp.param_groups[0]["lr"]) < 1e-12
    sd = m.aggregator.state_dict()
    for k, v in om.items():
        e = (sd[k].detach().cpu() - v.detach()).abs()
        assert float(e.max()) <= 2.0 * opt.lr * STEPS and float((e > 0.05 * opt.lr).float().mean()) <= 2e-3, k
    for k in ("points_embeding", "points_conf", "points_dir", "points_color"):
        e = (getattr(m.neural_points, k).detach().cpu().reshape(oa[k].shape) - oa[k].detach()).abs()
        assert float(e.max()) <= 2.0 * opt.plr * STEPS and float((e > 0.05 * opt.plr).float().mean()) <= 2e-3, k
    # test(): no gradient, same keys, full-R tensors; the checkpoint it writes restores into a second shell
    m.eval()
    m.set_input({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()})
    out = m.test()
    assert not out["coarse_raycolor"].requires_grad and set(m.get_current_visuals()) == {"gt_image", "coarse_raycolor", "queried_shading"}
    # the renderer's sync-free hit-ray index (stable sort of the flags, count from the counters) is what nonzero() would give
    assert torch.equal(out["_hit_index"], torch.nonzero(out["ray_mask"][0] > 0).squeeze(1))
    m.save_networks(3, {"total_steps": 3})
    opt2 = config.lego_train_opt(**CASES["small_k8"][0], gpu_ids=[0], checkpoints_dir=str(tmp_path), num_point=xyz.shape[0], is_train=0,
                                 resume_iter=3, resume_dir=str(tmp_path / "lego"), load_points=1)
    m2 = create_model(opt2)
    m2.setup(opt2)
    m2.set_input({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()})
    out2 = m2.test()
    assert torch.equal(out2["coarse_raycolor"], out["coarse_raycolor"]) and torch.equal(out2["ray_mask"], out["ray_mask"])


def test_probe_hole_on_the_device_matches_the_oracle(tmp_path):
    H = W = 800
    opt, m, xyz, attrs, mlp, _, seed = _scene("small_k8", tmp_path, is_train=0, prob_mul=0.4, prob_num_step=1)
    size = 44                                   # the 0.06-radius cloud covers ~17 px around the image centre: hits inside, misses around
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=30.0, x0=400 - size // 2, y0=400 - size // 2, size=size))
    points = dict(xyz=xyz, **attrs)
    with torch.no_grad():
        ref = pyref.render(opt, points, mlp, inp, nthreads=8)
        pr = pyref.probe_outputs(ref, points)
    hit = ref["ray_mask"][0] > 0
    assert 50 < int(hit.sum()) < hit.numel() - 50
    pix = inp["pixel_idx"][0].long()
    def to_map(compact, c):
        t = np.zeros((H, W, c), np.float32)
        t[pix[hit, 1].numpy(), pix[hit, 0].numpy()] = compact[0].numpy()
        return t
    maps = {k: to_map(v, v.shape[-1]) for k, v in pr.items()}
    col = np.zeros((H, W, 3), np.float32); col[pix[:, 1].numpy(), pix[:, 0].numpy()] = pyref.fill_invalid(ref, inp)["coarse_raycolor"][0].numpy()
    rm = np.zeros((H, W), np.float32); rm[pix[:, 1].numpy(), pix[:, 0].numpy()] = hit.float().numpy()
    edge = np.zeros((H, W), bool); edge[pix[:, 1].numpy(), pix[:, 0].numpy()] = True
    gt = np.zeros((H, W, 3), np.float32); gt[pix[:, 1].numpy(), pix[:, 0].numpy()] = inp["gt_image"][0].numpy()
    # a threshold in the widest gap of the oracle's opacities (random-init MLP: they are all ~1e-3), so that rounding cannot
    # flip a candidate
    op = np.sort(maps["ray_max_shading_opacity"][..., 0][rm > 0])
    gaps = np.diff(op)
    lo = len(op) // 4
    j = lo + int(np.argmax(gaps[lo:3 * len(op) // 4]))
    thresh = float(0.5 * (op[j] + op[j + 1]))
    assert gaps[j] > 1e-3 * thresh
    mask = pyref.probe_hole_mask(rm, maps["ray_max_shading_opacity"][..., 0], maps["ray_max_far_dist"][..., 0], col, gt,
                                 inp["bg_color"].numpy().reshape(1, 3), edge, thresh)
    assert 5 < mask.sum() < (rm > 0).sum()
    view = dict(inp, id=0)
    got = probe.probe_hole(m, [view], opt, H, W, test_steps=0, opacity_thresh=thresh, frame_ids=[0], chunk=700)
    xyz_a, emb_a, col_a, dir_a, conf_a = [t.cpu().numpy() for t in got]
    assert xyz_a.shape == (int(mask.sum()), 3) and emb_a.shape == (int(mask.sum()), 32)
    for a, k, s in ((xyz_a, "ray_max_sample_loc_w", 1.0), (emb_a, "shading_avg_embedding", 1.0), (col_a, "shading_avg_color", 1.0),
                    (dir_a, "shading_avg_dir", 1.0), (conf_a, "shading_avg_conf", 0.4)):
        assert float(np.abs(a - maps[k][mask] * np.float32(s)).max()) <= 1e-4, k
    assert opt.prob == 0 and opt.no_loss == 0
    # grow the proposals into the cloud: the next render hits at least as many rays, parameters and optimizers are rebuilt
    m.opt.is_train = 1
    m.setup_optimizer(opt)
    n0 = m.neural_points.xyz.shape[0]
    m.clean_optimizer_scheduler()
    m.grow_points(*got)
    m.setup_optimizer(opt); m.init_scheduler(5, opt)
    assert m.neural_points.xyz.shape[0] == n0 + int(mask.sum()) and any(p is m.neural_points.points_color for p in m.neural_params)
    m.opt.is_train = 0
    m.set_input({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()})
    out = m.test()
    assert int((out["ray_mask"] > 0).sum()) >= int(hit.sum()) and bool(torch.isfinite(out["coarse_raycolor"]).all())


def test_evaluation_loop_through_the_shell_matches_the_oracle(tmp_path):
    """eval_loop.test_views (= test() of run/train_ft.py:252-414) on two views of 30x30 rays in row-major order."""
    from pointnerf_amd import eval_loop
    H = W = 800
    opt, m, xyz, attrs, mlp, _, seed = _scene("small_k8", tmp_path, is_train=0)
    size = 30
    points = dict(xyz=xyz, **attrs)
    views, refs = [], []
    for i, theta in enumerate((30.0, 95.0)):
        inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=theta, x0=400 - size // 2, y0=400 - size // 2, size=size))
        views.append(dict(inp, id=i))
        with torch.no_grad():
            ref = pyref.render(opt, points, mlp, inp, nthreads=8)
        pix = inp["pixel_idx"][0].long().numpy()
        canvas = np.zeros((H, W, 3), np.float32)
        canvas[pix[:, 1], pix[:, 0]] = pyref.fill_invalid(ref, inp)["coarse_raycolor"][0].numpy()
        refs.append(pyref.test_view_losses(canvas, inp["gt_image"][0].numpy(), pix, (ref["ray_mask"][0] > 0).numpy(), H, W))
    got = []
    psnr, avg = eval_loop.test_views(m, views, opt, H, W, test_num_step=1, chunk=400, on_view=lambda i, v: got.append(v["coarse_raycolor"].clone()))
    for k in refs[0]:
        want = float(np.mean([r[k] for r in refs]))
        assert abs(avg[k] - want) <= 1e-4 * max(1.0, abs(want)), (k, avg[k], want)
    assert abs(psnr - avg["coarse_raycolor_psnr"]) < 1e-9 and len(got) == 2 and got[0].shape == (H, W, 3)
    assert float(got[0].abs().sum()) > 0 and float(got[0][:300].abs().sum()) == 0


def test_training_step_on_a_batch_that_hits_nothing(tmp_path):
    """A training batch may see no geometry at all (random pixels of a sky region, a shard of a sharded batch): the shell's step -- fused colour loss over
    the dense ray colours, zero-one numerator of zero elements, backward, both Adams -- must run, report the reference's value for the masked item
    (0: base_rendering_model.py:545-551 takes the mean only when a ray hit), fill the image with the background and leave the network unchanged
    (every gradient is exactly zero; Adam on an all-zero gradient from zero moments moves nothing)."""
    opt, m, xyz, attrs, mlp, size, seed = _scene("small_k8", tmp_path, prob_freq=100, prob_num_step=1, ray_jitter=0.0)
    d = scenes.block_rays(theta_deg=30.0, x0=400 - size // 2, y0=400 - size // 2, size=size)
    d["raydir"] = d["raydir"] + np.array([9.0, 0.0, 0.0], dtype=np.float32)                 # every ray far off to the side of the cloud
    inp = pyref.to_torch_inputs(d)
    m.setup(opt, train_len=10)
    m.train()
    before = {k: v.detach().clone() for k, v in m.aggregator.state_dict().items()}
    emb0 = m.neural_points.points_embeding.detach().clone()
    for step in range(2):
        data = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
        data["id"] = torch.tensor([3])
        m.set_input(data)
        m.optimize_parameters(total_steps=step)
        losses = {k: float(v) for k, v in m.get_current_losses().items()}
        assert all(np.isfinite(v) for v in losses.values()), losses
        assert losses["ray_masked_coarse_raycolor"] == 0.0 and not bool((m.output["ray_mask"] > 0).any())
        R = inp["raydir"].shape[1]
        assert m.output["coarse_raycolor"].shape == (1, R, 3)
        assert float((m.output["coarse_raycolor"][0] - inp["bg_color"].to(DEV).reshape(1, 3)).abs().max()) == 0.0
        miss_ref = float(((inp["bg_color"].reshape(1, 3) - inp["gt_image"][0]) ** 2).sum() / 3.0)
        assert abs(losses["ray_miss_coarse_raycolor"] - miss_ref) <= 1e-5 * max(1.0, miss_ref)
    for k, v in m.aggregator.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert torch.equal(m.neural_points.points_embeding.detach(), emb0)
