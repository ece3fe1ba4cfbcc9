"""CPU tests of the model shell the run scripts drive (``create_model`` -> MvsPointsVolumetricModel) and of the probe-and-grow
host logic: loss items against the oracle's restatement of base_rendering_model.py:533-662, the missed-ray ranking,
schedulers, the checkpoint round trip and the hole-selection rule of run/train_ft.py:417-530.  No compute call of the HIP
library happens here (the network forward is replaced by hand-made outputs)."""
import numpy as np
import pytest
import torch

from oracle import pyref
from pointnerf_amd import config, eval_loop, probe, scenes
from pointnerf_amd.mvs_points_volumetric_model import create_model, get_scheduler
from pointnerf_amd.neural_points_volumetric_model import fill_invalid, PROBE_KEYS


def _opt(tmp_path, **kw):
    base = dict(gpu_ids=[], checkpoints_dir=str(tmp_path), name="run", resume_dir="", num_point=50, K=4, SR=8)
    base.update(kw)
    return config.lego_train_opt(**base)


def _points(n, seed=0):
    a = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, 32, seed).items()}
    return torch.from_numpy(scenes.chair_points(n, seed=seed)), a


def _model(tmp_path, n=50, **kw):
    opt = _opt(tmp_path, num_point=n, **kw)
    m = create_model(opt)
    xyz, a = _points(n)
    m.set_points(xyz, a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"])
    return opt, m


def _fake_outputs(R=40, hits=23, SR=8, K=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    mask = torch.zeros(1, R, dtype=torch.int8)
    mask[0, torch.randperm(R, generator=g)[:hits]] = 1
    raw = dict(ray_mask=mask, coarse_raycolor=torch.rand(1, hits, 3, generator=g).requires_grad_(True),
               coarse_point_opacity=torch.rand(1, hits, SR, generator=g), coarse_is_background=torch.rand(1, hits, 1, generator=g),
               queried_shading=torch.zeros(1, hits, 3), weight=torch.rand(1, hits, SR, K, generator=g),
               conf_coefficient=torch.rand(1, hits, SR, K, generator=g).requires_grad_(True))
    return raw, torch.rand(1, R, 3, generator=g), torch.tensor([[1.0, 1.0, 1.0]])


def test_create_model_builds_the_reference_shell(tmp_path):
    opt, m = _model(tmp_path)
    assert m.name() == "MvsPointsVolumetricModel" and m.model_names == ["ray_marching"]
    assert m.loss_names == ["total", "ray_masked_coarse_raycolor", "ray_miss_coarse_raycolor", "coarse_raycolor", "conf_coefficient"]
    assert m.visual_names == ["gt_image", "coarse_raycolor", "queried_shading"]
    # two Adam instances: MLP parameters at lr, neural_points.* at plr (mvs_points_volumetric_model.py:80-91); xyz has no grad
    assert len(m.optimizers) == 2 and m.optimizer.param_groups[0]["lr"] == opt.lr and m.neural_point_optimizer.param_groups[0]["lr"] == opt.plr
    assert sum(p.numel() for p in m.net_params) == 341764
    assert len(m.neural_params) == 5 and sum(p.requires_grad for p in m.neural_params) == 4
    with pytest.raises(NotImplementedError):
        create_model(config.lego_train_opt(gpu_ids=[], checkpoints_dir=str(tmp_path), mode=0))
    with pytest.raises(NotImplementedError):
        m.gen_points()


@pytest.mark.parametrize("hits", [23, 0, 40])
def test_loss_items_equal_the_reference_formulas(tmp_path, hits):
    opt, m = _model(tmp_path)
    raw, gt, bg = _fake_outputs(hits=hits)
    m.set_input(dict(gt_image=gt, bg_color=bg))
    m._raw = raw
    m.output = fill_invalid(raw, bg)
    m.compute_losses()
    # no hit ray at all: the reference's zero-one item would be the mean of an empty tensor (NaN); ours is 0 there
    zo = opt.zero_one_loss_items if hits else []
    ref_total, parts = pyref.compute_losses({k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in m.output.items()}, gt,
                                            opt.color_loss_items, opt.color_loss_weights, zo, opt.zero_one_loss_weights, opt.zero_epsilon)
    assert abs(float(m.loss_total) - float(ref_total)) <= 1e-6 * max(1.0, abs(float(ref_total)))
    for k, v in parts.items():
        assert abs(float(getattr(m, "loss_" + k)) - float(v)) <= 1e-5 * max(1.0, abs(float(v))), k
    assert set(m.get_current_losses()) == {"total", "ray_masked_coarse_raycolor", "ray_miss_coarse_raycolor", "coarse_raycolor", "conf_coefficient"}
    # gradients reach the renderer's compact output (weight 1 on the masked item only) and the confidences
    m.loss_total.backward()
    if hits:
        g = raw["coarse_raycolor"].grad
        assert torch.allclose(g, 2 * (raw["coarse_raycolor"].detach() - gt[0][raw["ray_mask"][0] > 0][None]) / (hits * 3), atol=1e-7)
        assert raw["conf_coefficient"].grad.abs().sum() > 0


def test_sparse_loss_and_weight_broadcast(tmp_path):
    opt, m = _model(tmp_path, sparse_loss_weight=0.5, color_loss_items="coarse_raycolor ray_masked_coarse_raycolor", color_loss_weights=[2.0])
    assert opt.color_loss_weights == [2.0, 2.0] and "sparse" in m.loss_names
    raw, gt, bg = _fake_outputs()
    m.set_input(dict(gt_image=gt, bg_color=bg))
    m._raw, m.output = raw, fill_invalid(raw, bg)
    ref_total, parts = pyref.compute_losses({k: v.detach() for k, v in m.output.items()}, gt, opt.color_loss_items, opt.color_loss_weights,
                                            opt.zero_one_loss_items, opt.zero_one_loss_weights, opt.zero_epsilon, sparse_loss_weight=0.5)
    m.compute_losses()
    assert abs(float(m.loss_total) - float(ref_total)) <= 1e-6 and abs(float(m.loss_sparse) - float(parts["sparse"])) <= 1e-6
    assert "weight" not in m.output and "conf_coefficient" not in m.output          # popped like :655-656
    with pytest.raises(ValueError):
        create_model(_opt(tmp_path, color_loss_weights=[1.0, 2.0]))


def test_ray_miss_ranking(tmp_path):
    opt, m = _model(tmp_path, prob_freq=100, prob_num_step=4)
    m.setup(opt, train_len=20)
    assert m.num_probe == 5 and m.top_ray_miss_ids.tolist() == [0, 1, 2, 3, 4, 5] and float(m.top_ray_miss_loss.sum()) == 0
    ids, losses = list(range(6)), [0.0] * 6
    rng = np.random.default_rng(0)
    for step, (vid, l) in enumerate(zip(rng.integers(0, 20, 60), rng.permutation(60) + 1.0)):
        m.input = dict(id=torch.tensor([int(vid)]))
        m.loss_ray_miss_coarse_raycolor = torch.tensor(float(l))
        m.update_rank_ray_miss(step)
        losses, ids = pyref.rank_ray_miss(int(vid), float(l), ids, losses)
        assert m.top_ray_miss_loss.tolist() == losses and m.top_ray_miss_ids.tolist() == ids
    # past the last probe tier nothing is ranked any more (:136)
    before = m.top_ray_miss_loss.clone()
    m.loss_ray_miss_coarse_raycolor = torch.tensor(1e9)
    m.update_rank_ray_miss(opt.prob_tiers[-1] + 1)
    assert torch.equal(before, m.top_ray_miss_loss)
    # prob_num_step == 1: one running maximum (:141-142)
    opt1, m1 = _model(tmp_path, prob_freq=100, prob_num_step=1)
    m1.setup(opt1, train_len=20)
    for l in (0.5, 0.2, 0.9):
        m1.loss_ray_miss_coarse_raycolor = torch.tensor(l)
        m1.update_rank_ray_miss(0)
    assert abs(float(m1.top_ray_miss_loss[0]) - 0.9) < 1e-7


def test_schedulers_follow_the_script_policy(tmp_path):
    opt, m = _model(tmp_path, lr_decay_iters=1000)
    m.setup(opt, train_len=None)
    assert len(m.schedulers) == 2
    for _ in range(250):
        m.update_learning_rate(opt=opt, total_steps=1)
    assert abs(m.optimizer.param_groups[0]["lr"] - opt.lr * 0.1 ** 0.25) < 1e-12
    assert abs(m.neural_point_optimizer.param_groups[0]["lr"] - opt.plr * 0.1 ** 0.25) < 1e-12
    # rebuilding after prune / grow fast-forwards fresh schedulers to the same learning rate (run/train_ft.py:836-840)
    m.clean_optimizer(); m.clean_scheduler()
    m.setup_optimizer(opt); m.init_scheduler(250, opt)
    assert abs(m.neural_point_optimizer.param_groups[0]["lr"] - opt.plr * 0.1 ** 0.25) < 1e-12
    lam = get_scheduler(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0), config.lego_train_opt(lr_policy="lambda", niter=2, niter_decay=3))
    assert [round(lam.lr_lambdas[0](i), 6) for i in (0, 2, 3, 6)] == [1.0, 1.0, 0.75, 0.0]
    with pytest.raises(NotImplementedError):
        get_scheduler(m.optimizer, config.lego_train_opt(lr_policy="cosine_annealing"))


def test_checkpoint_round_trip_and_default_conf(tmp_path):
    opt, m = _model(tmp_path, n=50)
    m.save_networks(300, {"total_steps": 300, "best_PSNR": 31.5})
    states = torch.load(str(tmp_path / "run" / "300_states.pth"))
    assert states == {"total_steps": 300, "best_PSNR": 31.5}
    sd = torch.load(str(tmp_path / "run" / "300_net_ray_marching.pth"))
    assert "neural_points.points_embeding" in sd and "aggregator.block1.0.weight" in sd and all(v.device.type == "cpu" for v in sd.values())
    # a fresh model whose checkpoint dir holds that file restores the cloud in create_network_models; resume_dir loads the MLP
    opt2 = _opt(tmp_path, num_point=50, resume_iter=300, resume_dir=str(tmp_path / "run"), load_points=1)
    m2 = create_model(opt2)
    m2.setup(opt2)
    for k, v in m.net_ray_marching.state_dict().items():
        assert torch.equal(v, m2.net_ray_marching.state_dict()[k]), k
    # resume keeps every scheduler attached to the optimizer that is actually stepped (the learning rate must keep decaying)
    assert len(m2.schedulers) == len(m2.optimizers) == 2 and all(s.optimizer is o for s, o in zip(m2.schedulers, m2.optimizers))
    lr0 = [o.param_groups[0]["lr"] for o in m2.optimizers]
    for _ in range(2000):
        m2.update_learning_rate(verbose=False)
    assert all(o.param_groups[0]["lr"] < l for o, l in zip(m2.optimizers, lr0))
    # "best" checkpoint without stored confidences: default_conf fills them (mvs_points_volumetric_model.py:318-320); a
    # checkpoint with a different point count replaces the parameters and the optimizer is rebuilt on the new ones
    sd_best = {k: v for k, v in sd.items() if k != "neural_points.points_conf"}
    xyz, a = _points(70, seed=3)
    sd_best.update({"neural_points.xyz": xyz, "neural_points.points_embeding": a["points_embeding"], "neural_points.points_dir": a["points_dir"],
                    "neural_points.points_color": a["points_color"]})
    torch.save(sd_best, str(tmp_path / "run" / "best_net_ray_marching.pth"))
    opt3 = _opt(tmp_path, num_point=70, resume_iter="none", resume_dir=str(tmp_path / "run"))
    m3 = create_model(opt3)
    x0, a0 = _points(70, seed=5)
    m3.set_points(x0, a0["points_embeding"], points_color=a0["points_color"], points_dir=a0["points_dir"], points_conf=a0["points_conf"])
    m3.load_networks("best")
    assert torch.equal(m3.neural_points.xyz.data, xyz) and float((m3.neural_points.points_conf - opt3.default_conf).abs().max()) == 0
    assert any(p is m3.neural_points.points_embeding for p in m3.neural_params)
    # the parameters were replaced: optimizers rebuilt, and schedulers (if any) rebuilt on them
    m3.init_scheduler(10, opt3)
    m3.load_networks("best")
    assert all(s.optimizer is o for s, o in zip(m3.schedulers, m3.optimizers))


def test_fill_invalid_scatters_probe_outputs_and_bg_ray():
    raw, gt, bg = _fake_outputs(R=30, hits=11)
    hits = 11
    g = torch.Generator().manual_seed(9)
    shapes = dict(ray_max_sample_loc_w=3, ray_max_shading_opacity=1, shading_avg_color=3, shading_avg_dir=3, shading_avg_conf=1,
                  shading_avg_embedding=32, ray_max_far_dist=1)
    assert set(shapes) == set(PROBE_KEYS)
    for k, c in shapes.items():
        raw[k] = torch.rand(1, hits, c, generator=g)
    full = fill_invalid(raw, bg, prob=1)
    sel = raw["ray_mask"][0] > 0
    for k, c in shapes.items():
        assert full[k].shape == (1, 30, c) and torch.equal(full[k][0, sel], raw[k][0]) and float(full[k][0, ~sel].abs().sum()) == 0
    assert fill_invalid(raw, bg, prob=0)["ray_max_far_dist"].shape == (1, hits, 1)
    bg_ray = torch.rand(1, 30, 3, generator=g)
    col = fill_invalid(raw, bg, bg_ray=bg_ray)["coarse_raycolor"]
    assert torch.allclose(col[0, ~sel], bg_ray[0, ~sel])
    assert torch.allclose(col[0, sel], raw["coarse_raycolor"][0] + raw["coarse_is_background"][0] * bg_ray[0, sel])


@pytest.mark.parametrize("far_thresh", [-1.0, 0.01])
def test_hole_mask_equals_the_index_loop_restatement(far_thresh):
    H, W = 23, 31
    rng = np.random.default_rng(4)
    ray_mask = (rng.random((H, W)) > 0.35)
    ray_mask[:, 0] = False; ray_mask[0, :] = False                      # misses on the border exercise the clamp
    opacity = rng.random((H, W)).astype(np.float32)
    far = (rng.random((H, W)) * 0.02).astype(np.float32)
    gt = rng.random((H, W, 3)).astype(np.float32)
    gt[rng.random((H, W)) < 0.3] = 1.0                                  # background-coloured pixels are not holes
    col = (gt + rng.normal(0, 0.06, (H, W, 3))).astype(np.float32)
    edge = rng.random((H, W)) > 0.1
    gt[~edge] = 0
    bg = np.ones((1, 3), np.float32)
    ref = pyref.probe_hole_mask(ray_mask.astype(np.float32), opacity, far, col, gt, bg, edge, 0.4, far_thresh)
    t = torch.from_numpy
    maps = dict(ray_mask=t(ray_mask.astype(np.float32))[..., None], ray_max_shading_opacity=t(opacity)[..., None],
                ray_max_far_dist=t(far)[..., None], coarse_raycolor=t(col))
    got = probe.hole_mask(maps, t(gt), t(bg), t(edge), 0.4, far_thresh)
    assert got.dtype == torch.bool and np.array_equal(got.numpy(), ref) and 0 < ref.sum() < ray_mask.sum()
    # bloat_inds (kept for callers) marks the same pixels as the pooling
    miss = t((~ray_mask) & (np.linalg.norm(gt - bg, axis=-1) > 0.002) & edge)
    b = probe.bloat_inds(miss.nonzero(), 1, H, W)
    m2 = torch.zeros(H, W); m2[b[:, 0], b[:, 1]] = 1
    pooled = torch.nn.functional.max_pool2d(miss[None, None].float(), 3, 1, 1)[0, 0]
    assert torch.equal(m2, pooled)


class _FakeModel:
    """Stands in for the model shell: ``test()`` returns hand-made probe outputs for the rays of the current chunk."""

    def __init__(self, opt, H, W, seed):
        self.opt, self.device, self.H, self.W = opt, torch.device("cpu"), H, W
        g = torch.Generator().manual_seed(seed)
        self.hit = torch.rand(H, W, generator=g) > 0.4
        self.maps = dict(ray_max_sample_loc_w=torch.rand(H, W, 3, generator=g), ray_max_far_dist=torch.rand(H, W, 1, generator=g),
                         ray_max_shading_opacity=torch.rand(H, W, 1, generator=g), shading_avg_color=torch.rand(H, W, 3, generator=g),
                         shading_avg_dir=torch.rand(H, W, 3, generator=g), shading_avg_conf=torch.rand(H, W, 1, generator=g),
                         shading_avg_embedding=torch.rand(H, W, 32, generator=g), coarse_raycolor=torch.rand(H, W, 3, generator=g))
        self.calls, self.seen = 0, []

    def set_input(self, d):
        self.input = d

    def test(self):
        self.calls += 1
        self.seen.append((self.opt.prob, self.opt.no_loss, tuple(np.asarray(self.opt.query_size).tolist())))
        p = self.input["pixel_idx"][0].long()
        hit = self.hit[p[:, 1], p[:, 0]]
        out = {k: (v[p[:, 1], p[:, 0]] * hit[:, None])[None] for k, v in self.maps.items()}
        out["ray_mask"] = hit[None].to(torch.int8)
        if not bool(hit.any()):
            out = dict(coarse_raycolor=out["coarse_raycolor"], ray_mask=out["ray_mask"])
        return out


def test_probe_hole_accumulates_over_views_like_the_reference():
    H, W = 12, 16
    opt = config.lego_train_opt(prob_kernel_size=[5, 5, 5, 7, 7, 7], prob_tiers=[100, 200], prob_mul=0.5, prob_num_step=1)
    model = _FakeModel(opt, H, W, seed=1)
    opt.query_size = [3, 3, 3]
    py, px = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pix = torch.stack([px, py], -1)[None].float()                         # [1,H,W,2]
    g = torch.Generator().manual_seed(2)
    views = [dict(raydir=torch.zeros(1, H * W, 3), pixel_idx=pix, gt_image=torch.rand(1, H * W, 3, generator=g), bg_color=torch.ones(1, 3), id=i)
             for i in range(3)]
    xyz, emb, color, dirs, conf = probe.probe_hole(model, views, opt, H, W, test_steps=150, opacity_thresh=0.3, frame_ids=[2, 0], chunk=50)
    assert model.calls == 2 * ((H * W + 49) // 50)
    assert set(model.seen) == {(1, 1, (7, 7, 7))}                          # tier 1 at step 150; prob / no_loss set while probing
    assert opt.prob == 0 and opt.no_loss == 0 and list(opt.query_size) == [3, 3, 3]      # ...and restored afterwards
    exp = dict(xyz=[], emb=[], color=[], dir=[])
    conf_ref = np.zeros((0, 1), np.float32)
    for i in (2, 0):
        gt = views[i]["gt_image"].reshape(H, W, 3).numpy()
        m = pyref.probe_hole_mask(model.hit.float().numpy(), (model.maps["ray_max_shading_opacity"][..., 0] * model.hit).numpy(),
                                  model.maps["ray_max_far_dist"][..., 0].numpy(), model.maps["coarse_raycolor"].numpy(), gt,
                                  np.ones((1, 3), np.float32), np.ones((H, W), bool), 0.3)
        exp["xyz"].append(model.maps["ray_max_sample_loc_w"].numpy()[m]); exp["emb"].append(model.maps["shading_avg_embedding"].numpy()[m])
        exp["color"].append(model.maps["shading_avg_color"].numpy()[m]); exp["dir"].append(model.maps["shading_avg_dir"].numpy()[m])
        conf_ref = np.concatenate([conf_ref, model.maps["shading_avg_conf"].numpy()[m]], 0) * np.float32(0.5)    # the reference's cumulative scaling
    assert len(xyz) > 0 and np.array_equal(xyz.numpy(), np.concatenate(exp["xyz"])) and np.array_equal(emb.numpy(), np.concatenate(exp["emb"]))
    assert np.array_equal(color.numpy(), np.concatenate(exp["color"])) and np.array_equal(dirs.numpy(), np.concatenate(exp["dir"]))
    assert np.allclose(conf.numpy(), conf_ref, atol=1e-7)


def test_frame_selection_and_prune_rebuild(tmp_path):
    opt, m = _model(tmp_path, n=60, prob_freq=100, prob_num_step=4, prune_iter=50, prune_thresh=0.5)
    m.setup(opt, train_len=20)
    m.top_ray_miss_loss = torch.tensor([0.9, 0.5, 0.2, 0.0, 0.0, 0.0])
    m.top_ray_miss_ids = torch.tensor([7, 3, 11, 0, 1, 2], dtype=torch.int32)
    ids, top = probe.select_probe_frames(m, 20, opt)
    assert top and ids == [7, 3, 11]
    opt.prob_top = 0
    ids, top = probe.select_probe_frames(m, 20, opt)
    assert not top and sorted(ids) == [0, 1, 2, 3, 4]
    # prune at a prune_iter multiple: points below the confidence threshold go, both optimizers are rebuilt on the new
    # parameters and the schedulers are fast-forwarded
    keep = int((m.neural_points.points_conf[0, :, 0] >= 0.5).sum())
    old = m.neural_point_optimizer
    opt.prob_freq = 0
    assert probe.prune_and_grow_step(m, [], opt, total_steps=100, height=4, width=4) == 0
    assert m.neural_points.xyz.shape[0] == keep < 60 and m.neural_point_optimizer is not old
    assert any(p is m.neural_points.points_conf for p in m.neural_params) and len(m.schedulers) == 2
    assert m.schedulers[0].last_epoch == 100
    assert probe.prune_and_grow_step(m, [], opt, total_steps=101, height=4, width=4) == 0 and m.neural_points.xyz.shape[0] == keep


class _FakeShell:
    """Stands in for the model shell in the evaluation loop: ``test()`` colours the rays of the current chunk from a map."""

    def __init__(self, H, W, seed):
        g = torch.Generator().manual_seed(seed)
        self.device, self.img, self.hit = torch.device("cpu"), torch.rand(H, W, 3, generator=g), torch.rand(H, W, generator=g) > 0.3
        self.chunks = 0

    def eval(self):
        pass

    def set_input(self, d):
        self.input = d

    def test(self):
        self.chunks += 1
        p = self.input["pixel_idx"][0].long()
        self.output = dict(coarse_raycolor=self.img[p[:, 1], p[:, 0]][None], ray_mask=self.hit[p[:, 1], p[:, 0]][None].to(torch.int8))
        return self.output

    def get_current_visuals(self, data=None):
        return dict(gt_image=self.input["gt_image"], coarse_raycolor=self.output["coarse_raycolor"], queried_shading=None)


def test_evaluation_loop_scores_views_like_the_reference():
    H, W = 20, 24
    opt = config.lego_train_opt()
    y0, x0, h, w = 3, 5, 12, 14                                      # the view provides rays for a sub-rectangle only
    py, px = torch.meshgrid(torch.arange(y0, y0 + h), torch.arange(x0, x0 + w), indexing="ij")
    pix = torch.stack([px, py], -1)[None].float()
    g = torch.Generator().manual_seed(3)
    views = [dict(raydir=torch.zeros(1, h * w, 3), pixel_idx=pix, gt_image=torch.rand(1, h * w, 3, generator=g), id=i) for i in range(5)]
    shell = _FakeShell(H, W, 1)
    seen = []
    psnr, avg = eval_loop.test_views(shell, views, opt, H, W, test_num_step=2, chunk=50, on_view=lambda i, v: seen.append((i, sorted(v))))
    assert [i for i, _ in seen] == [0, 2, 4] and seen[0][1] == ["coarse_raycolor", "gt_image"] and shell.chunks == 3 * 4
    canvas = np.zeros((H, W, 3), np.float32)
    canvas[y0:y0 + h, x0:x0 + w] = shell.img[y0:y0 + h, x0:x0 + w].numpy()
    p = pix[0].reshape(-1, 2).long().numpy()
    refs = [pyref.test_view_losses(canvas, views[i]["gt_image"][0].numpy(), p, shell.hit[p[:, 1], p[:, 0]].numpy(), H, W) for i in (0, 2, 4)]
    for k in ("coarse_raycolor", "ray_masked_coarse_raycolor", "coarse_raycolor_psnr", "ray_masked_coarse_raycolor_psnr"):
        assert abs(avg[k] - np.mean([r[k] for r in refs])) <= 1e-5 * max(1.0, abs(np.mean([r[k] for r in refs]))), k
    assert abs(psnr - avg["coarse_raycolor_psnr"]) < 1e-12           # the first item of test_color_loss_items is what test() returns


def test_compute_losses_on_the_fused_colour_form_equals_the_compacted_form(tmp_path, monkeypatch):
    """MvsPointsVolumetricModel.compute_losses with the renderer's dense results (`_dense_color`: the form a device training step hands over since round 5)
    against the same data in the compacted form: every loss item of the lego script (ray_masked 1.0, ray_miss 0.0, full image 0.0) and the total agree,
    and only the ray_masked item carries a gradient to the ray colours.  The fused pass itself (ops.ColorLossRays, a HIP kernel) is replaced by its
    torch statement here; tests/test_gpu_backward.py compares the kernel with that statement on the device."""
    import torch
    from pointnerf_amd import config, ops
    from pointnerf_amd.mvs_points_volumetric_model import create_model
    from pointnerf_amd.neural_points_volumetric_model import fill_invalid
    monkeypatch.setattr(ops, "color_loss_sum_rays", lambda color, gt, ray_hit: (((color - gt) ** 2) * (ray_hit > 0)[:, None]).sum())
    g = torch.Generator().manual_seed(9)
    R, SR = 41, 8
    hit = torch.rand(R, generator=g) < 0.55
    idx = torch.nonzero(hit).squeeze(1)
    gt, bg = torch.rand(1, R, 3, generator=g), torch.ones(1, 3)
    opacity, bg_trans = torch.rand(R, SR, generator=g), torch.rand(R, generator=g)
    res = {}
    for form in ("compacted", "dense"):
        color = torch.rand(R, 3, generator=torch.Generator().manual_seed(10)).requires_grad_(True)
        opt = config.lego_train_opt(gpu_ids=[], checkpoints_dir=str(tmp_path), name="run_" + form, num_point=0, K=4, SR=SR)
        m = create_model(opt)
        m.set_input(dict(gt_image=gt, bg_color=bg))
        if form == "compacted":
            raw = dict(ray_mask=hit.to(torch.int8)[None], _hit_index=idx, coarse_raycolor=color[idx][None], coarse_point_opacity=opacity[idx][None],
                       coarse_is_background=bg_trans[idx][None, :, None], queried_shading=torch.zeros(1, idx.numel(), 3))
        else:
            raw = dict(ray_mask=hit.to(torch.int8)[None], _dense_color=(color, hit.to(torch.int32), int(hit.sum())), _dense_aux=(opacity, bg_trans))
        m._raw, m.output = raw, fill_invalid(raw, bg)
        m.compute_losses()
        m.loss_total.backward()
        res[form] = ({k: float(v) for k, v in m.get_current_losses().items()}, color.grad.clone())
    (la, ga), (lb, gb) = res["compacted"], res["dense"]
    assert set(la) == set(lb)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(la[k])), (k, la[k], lb[k])
    assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-9) and float(gb[~hit].abs().max()) == 0.0 and float(gb[hit].abs().max()) > 0.0
