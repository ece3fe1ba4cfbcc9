"""The REFERENCE's unmodified Python on top of the ``models`` overlay (pointnerf_amd/overlay): its option parser on the lego
script's own command line, its model factory and its model shell (MvsPointsVolumetricModel -> NeuralPointsVolumetricModel ->
BaseRenderingModel -> BaseModel: set_points / setup / set_input / optimize_parameters / test / state_dict), with the kernels
running under the host emulation.  Needs the reference checkout (/root/reference: the authoring container); skipped elsewhere."""
import os
import sys

import numpy as np
import pytest
import torch

import ref_overlay_util as U

pytestmark = pytest.mark.skipif(not U.available(), reason="the reference checkout is not present on this machine")


@pytest.fixture(scope="module")
def ref_model():
    from emu_util import emu_backend
    from pointnerf_amd import scenes
    opt = U.parse_options(["--gpu_ids", "-1", "--num_point", "1200", "--checkpoints_dir", "/tmp/pnerf_overlay_ckpt", "--resume_dir", "/tmp/pnerf_overlay_none",
                           "--SR", "12", "--K", "8", "--P", "24", "--max_o", "50000", "--ranges", "-0.3", "-0.3", "-0.3", "0.3", "0.3", "0.3",
                           "--random_sample_size", "5"])
    opt.mode = 2                       # run/train_ft.py:629: per-scene optimisation, no MVSNet
    opt.is_train = True
    from models import create_model   # the reference's factory, through the overlay
    with emu_backend():
        model = create_model(opt)
        # with --gpu_ids 0 the reference wraps the network (neural_points_volumetric_model.py:165-168) and selects the point parameters by
        # the "module." prefix the wrapper adds (:189-190); no GPU here, so the (pass-through) wrapper is put on by hand
        model.net_ray_marching = torch.nn.DataParallel(model.net_ray_marching)
        n = 1200
        xyz = torch.from_numpy(scenes.chair_points(n, seed=5, radius=0.06))
        a = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, 32, 5).items()}
        # run/train_ft.py:756-760
        model.set_points(xyz, a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"])
        model.setup(opt, train_len=100)
        model.train()
        yield opt, model, xyz, a


def test_reference_modules_resolve_through_the_overlay(ref_model):
    opt, model, xyz, a = ref_model
    import models
    import models.mvs_points_volumetric_model as shell
    import models.neural_points_volumetric_model as nv
    import models.neural_points.point_query as pq
    import models.rendering.diff_ray_marching as drm
    assert shell.__file__.startswith(U.REF) and type(model).__module__ == "models.mvs_points_volumetric_model"     # the reference's own shell
    assert nv.NeuralPointsVolumetricModel.__module__ == "models._reference_neural_points_volumetric_model"
    assert pq.lighting_fast_querier.__module__ == "pointnerf_amd.point_query" and drm.ray_march.__module__ == "pointnerf_amd.diff_ray_marching"
    assert callable(drm.find_ray_generation_method)                                    # the names we do not replace are the reference's
    assert type(model.net_ray_marching.module).__module__ == "pointnerf_amd.neural_points_volumetric_model"
    # the options are the script's own (dev_scripts/w_n360/lego_cuda.sh)
    assert opt.model == "mvs_points_volumetric" and opt.vsize == [0.004, 0.004, 0.004] and opt.agg_dist_pers == 20 and opt.kernel_size == [3, 3, 3]
    # checkpoint keys of the network the shell saves (base_model.py:85-120)
    keys = set(model.net_ray_marching.module.state_dict().keys())
    for k in ("neural_points.xyz", "neural_points.points_embeding", "neural_points.points_conf", "neural_points.points_dir", "neural_points.points_color",
              "aggregator.block1.0.weight", "aggregator.block3.2.bias", "aggregator.alpha_branch.0.weight", "aggregator.color_branch.6.bias"):
        assert k in keys, k
    # two Adam instances over (MLP, point) parameters, as the reference's setup_optimizer builds them (the overlay does not touch it)
    assert len(model.optimizers) == 2 and all(isinstance(o, torch.optim.Adam) for o in model.optimizers)
    assert len(model.neural_params) == 5 and len(model.net_params) == 18        # xyz (frozen: xyz_grad 0), embedding, conf, dir, colour | 9 Linear layers


def test_reference_shell_trains_a_step_on_the_emulated_kernels(ref_model):
    from emu_util import emu_backend
    from pointnerf_amd import scenes
    from oracle import pyref
    opt, model, xyz, a = ref_model
    d = scenes.block_rays(theta_deg=55.0, x0=398, y0=398, size=5)
    data = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
    data["id"] = torch.tensor([3])      # the dataset's view id (data/nerf_synth360_ft_dataset.py: ranked by update_rank_ray_miss)
    opt.ray_jitter = 0.0               # ours: pins the in-kernel jitter so that the step is comparable with the oracle
    sd = {k: v.detach().clone() for k, v in model.net_ray_marching.module.state_dict().items()}
    mlp = {k[len("aggregator."):]: v for k, v in sd.items() if k.startswith("aggregator.")}
    # the cloud as the model holds it (set_points replaced the confidences by --default_conf 0.15, neural_points.py:472-473)
    pts = dict(xyz=sd["neural_points.xyz"], **{k: sd["neural_points." + k] for k in ("points_embeding", "points_conf", "points_dir", "points_color")})
    assert float((pts["points_conf"] - opt.default_conf).abs().max()) == 0.0
    with emu_backend():
        model.set_input(data)                                   # run/train_ft.py:927
        model.optimize_parameters(total_steps=1)                # :937  forward, the reference's compute_losses, backward, two Adam steps
        losses = model.get_current_losses()
    oracle_opt = opt                   # the oracle reads the very namespace the reference's parser produced
    inp = pyref.to_torch_inputs(d)
    ref = pyref.render(oracle_opt, pts, mlp, inp)
    want = float(pyref.training_loss(oracle_opt, ref, inp))
    got = float(losses["total"])
    print("loss through the reference's shell on our kernels: %.8f, oracle: %.8f" % (got, want))
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want))
    # the step changed the parameters (both optimizers ran)
    assert not torch.equal(model.net_ray_marching.module.state_dict()["aggregator.block1.0.weight"], mlp["block1.0.weight"])


def test_reference_forward_body_runs_module_by_module_on_the_overlay():
    """PNERF_OVERLAY_FUSED=0: the network is the reference's own NeuralPointsRayMarching (its forward body, :252-364), calling the
    overlay's NeuralPoints.forward (14-tuple), PointAggregator.forward and ray_march one by one; same loss as the oracle"""
    import json
    import subprocess
    env = dict(os.environ, PNERF_OVERLAY_FUSED="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "overlay_step.py")], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(out)
    assert out["network"] == "models._reference_neural_points_volumetric_model" and out["updated"]
    assert abs(out["loss"] - out["oracle"]) <= 1e-5 * max(1.0, abs(out["oracle"]))


def test_overlay_serves_extract_2d_and_query_embedding_of_the_reference_class():
    """models/mvs/mvs_points_model.py through the overlay is the reference's module (gen_points, the networks: its own code) whose
    extract_2d / query_embedding are the library's."""
    U.install()
    import models.mvs.mvs_points_model as M
    from pointnerf_amd import mvs_points_model as A
    assert M._ref.__file__.startswith(U.REF)
    assert M.MvsPointsModel is M._ref.MvsPointsModel and M.MvsPointsModel.__module__ == "models.mvs._reference_mvs_points_model"
    assert M.MvsPointsModel.extract_2d is M._extract_2d and M.MvsPointsModel.query_embedding is M._query_embedding
    assert M._ref_extract_2d.__module__ == "models.mvs._reference_mvs_points_model" and M._ref_extract_2d is not A.MvsPointsModel.extract_2d
    assert "gen_points" in vars(M._ref.MvsPointsModel) and M.MvsPointsModel.gen_points is not A.MvsPointsModel.gen_points
    import models.mvs_points_volumetric_model as shell
    assert shell.MvsPointsModel is M.MvsPointsModel


def test_overlay_keeps_the_reference_methods_where_gradients_are_wanted(monkeypatch):
    """the reference's extract_2d / query_embedding are differentiable (gen_points -> query_embedding carries the point features'
    gradient to FeatureNet in the feed-forward training scripts, models/mvs/mvs_points_model.py:370); the HIP versions return leaves.  Through the
    overlay a call whose feature maps / positions / confidences carry a gradient while autograd records runs the REFERENCE'S method; the same call
    under no_grad, or with plain tensors, runs the library's."""
    import torch
    U.install()
    import models.mvs.mvs_points_model as M
    from pointnerf_amd import mvs_points_model as A
    calls = []
    monkeypatch.setattr(M, "_ref_extract_2d", lambda self, *a, **k: calls.append("ref_extract") or "R")
    monkeypatch.setattr(M, "_ref_query_embedding", lambda self, *a, **k: calls.append("ref_query") or "R")
    monkeypatch.setattr(A.MvsPointsModel, "extract_2d", lambda self, *a, **k: calls.append("hip_extract") or "H")
    monkeypatch.setattr(A.MvsPointsModel, "query_embedding", lambda self, *a, **k: calls.append("hip_query") or "H")
    me = object.__new__(M.MvsPointsModel)
    feats = [torch.zeros(3, 3, 4, 4), torch.zeros(3, 8, 4, 4, requires_grad=True)]
    plain = [f.detach() for f in feats]
    xyz, conf = torch.zeros(1, 5, 3), torch.ones(1, 5, 1)
    ex = lambda f, x: M.MvsPointsModel.extract_2d(me, f, [0], [0, 1], None, None, None, x, 4, 4, cam_vid=0)
    qe = lambda f, x, c: M.MvsPointsModel.query_embedding(me, (4, 4), x, c, f, None, None, None, 0, pointdir_w=True)
    assert ex(feats, xyz) == "R" and ex(plain, xyz) == "H" and ex(plain, xyz.clone().requires_grad_(True)) == "R"
    with torch.no_grad():
        assert ex(feats, xyz) == "H" and qe(feats, xyz, conf) == "H"
    assert qe(feats, xyz, conf) == "R" and qe(plain, xyz, conf) == "H" and qe(plain, xyz, conf.clone().requires_grad_(True)) == "R"
    assert qe(plain, xyz, None) == "H"
    assert calls == ["ref_extract", "hip_extract", "ref_extract", "hip_extract", "hip_query", "ref_query", "hip_query", "ref_query", "hip_query"]
