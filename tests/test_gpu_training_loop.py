"""The loop of run/train_ft.py around the hot path, end to end on the device with the model shell:
train steps (set_input / optimize_parameters / update_learning_rate), the prune step, the probe-and-grow step over the training
views (probe.prune_and_grow_step), the evaluation loop (eval_loop.test_views) and a checkpoint round trip -- every caller of the
path that SURVEY.md 8f lists, wired the way the reference wires them (run/train_ft.py:756-765, 834-880, 252-414)."""
import numpy as np
import pytest
import torch

from cases import CASES
from oracle import pyref
from pointnerf_amd import config, eval_loop, probe, scenes
from pointnerf_amd.mvs_points_volumetric_model import create_model

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_train_prune_probe_grow_eval_checkpoint(tmp_path):
    ov, n, _, seed = CASES["small_k8"]
    opt = config.lego_train_opt(**ov, gpu_ids=[0], checkpoints_dir=str(tmp_path), num_point=n, default_conf=-1.0, ray_jitter=0.0,
                                prune_iter=3, prune_thresh=0.2, prune_max_iter=100, prob_freq=4, prob_num_step=1, prob_thresh=0.0,
                                prob_mul=0.5, maximum_step=100)
    torch.manual_seed(0)
    m = create_model(opt)
    xyz = torch.from_numpy(scenes.chair_points(n, seed=seed, radius=0.06))
    a = {k: torch.from_numpy(v).to(DEV) for k, v in scenes.point_attributes(n, opt.point_features_dim, seed).items()}
    m.set_points(xyz.to(DEV), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"])
    size = 40                                   # 40x40 rays around the cloud's silhouette: hits inside, misses (= "holes": gt is not background) around
    views = [dict(pyref.to_torch_inputs(scenes.block_rays(theta_deg=th, x0=400 - size // 2, y0=400 - size // 2, size=size)), id=torch.tensor([i]))
             for i, th in enumerate((20.0, 60.0, 100.0))]
    m.setup(opt, train_len=len(views))
    m.train()
    H = W = 800
    losses, n_points, added = [], [], []
    for step in range(1, 9):
        data = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in views[step % 3].items()}
        m.set_input(data)
        m.optimize_parameters(total_steps=step)
        m.update_learning_rate(opt=opt, total_steps=step)
        losses.append(float(m.get_current_losses()["total"]))
        added.append(probe.prune_and_grow_step(m, views, opt, total_steps=step, height=H, width=W))
        n_points.append(int(m.neural_points.xyz.shape[0]))
    assert all(np.isfinite(losses)), losses
    # step 3 and 6: prune (confidences below 0.2 go); step 4 and 8: probe all views and grow the proposals into the cloud
    assert n_points[2] < n and added[2] == 0                     # pruned at step 3
    assert added[3] > 0 and n_points[3] == n_points[2] + added[3]  # grown at step 4
    assert added[0] == added[1] == added[4] == 0
    # after every rebuild the optimizers hold the new parameters and the schedulers are where the loop is
    assert any(p is m.neural_points.points_embeding for p in m.neural_params) and m.schedulers[0].last_epoch == 8
    assert m.neural_points.points_conf.shape[1] == n_points[-1] and m.opt.prob == 0 and m.opt.is_train == 1
    # the step after a grow trains on the grown cloud (grid rebuilt from the new parameters) and stays finite
    assert np.isfinite(losses[4]) and np.isfinite(losses[-1])
    # evaluation loop over the views + checkpoint round trip
    m.opt.is_train = 0
    psnr, avg = eval_loop.test_views(m, views, opt, H, W, test_num_step=1)
    assert np.isfinite(psnr) and set(avg) >= {"coarse_raycolor", "coarse_raycolor_psnr", "ray_masked_coarse_raycolor"}
    m.save_networks(8, {"total_steps": 8})
    opt2 = config.lego_train_opt(**ov, gpu_ids=[0], checkpoints_dir=str(tmp_path), num_point=n, is_train=0, resume_iter=8,
                                 resume_dir=str(tmp_path / "lego"), load_points=1, default_conf=-1.0)
    m2 = create_model(opt2)
    m2.setup(opt2)
    assert m2.neural_points.xyz.shape[0] == n_points[-1]
    psnr2, _ = eval_loop.test_views(m2, views, opt2, H, W, test_num_step=1)
    assert abs(psnr2 - psnr) < 1e-4
