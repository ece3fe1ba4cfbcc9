"""The REFERENCE's unmodified Python (option parser on lego_cuda.sh's command line, models.create_model, the model shell
MvsPointsVolumetricModel -> ... -> BaseModel) on top of the ``models`` overlay of pointnerf_amd, with the kernels running in
libpnerf_hip.so ON THE DEVICE: set_points / setup / set_input / optimize_parameters / test, loss and rendered colours against the
oracle.  tests/test_reference_overlay.py does the same on the host emulator; this is the run on an MI355X.

Needs a reference checkout next to the repo: skipped unless POINTNERF_REFERENCE points at one (the GPU box has none by default;
tools/gpu_overlay_run.sh ships one for the duration of a gpurun call and keeps the log under profiles/)."""
import os

import numpy as np
import pytest
import torch

import ref_overlay_util as U

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (os.environ.get("POINTNERF_REFERENCE") and U.available()),
                                                 reason="POINTNERF_REFERENCE does not point at a reference checkout")]


@pytest.fixture(scope="module")
def ref_model():
    from pointnerf_amd import scenes
    opt = U.parse_options(["--gpu_ids", "0", "--num_point", "1200", "--checkpoints_dir", "/tmp/pnerf_overlay_ckpt", "--resume_dir", "/tmp/pnerf_overlay_none",
                           "--SR", "12", "--K", "8", "--P", "24", "--max_o", "50000", "--ranges", "-0.3", "-0.3", "-0.3", "0.3", "0.3", "0.3",
                           "--random_sample_size", "5"])
    opt.mode = 2                       # run/train_ft.py:629: per-scene optimisation, no MVSNet
    opt.is_train = True
    from models import create_model   # the reference's factory, through the overlay
    model = create_model(opt)
    n = 1200
    xyz = torch.from_numpy(scenes.chair_points(n, seed=5, radius=0.06)).cuda()
    a = {k: torch.from_numpy(v).cuda() for k, v in scenes.point_attributes(n, 32, 5).items()}
    model.set_points(xyz, a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"])   # run/train_ft.py:756-760
    model.setup(opt, train_len=100)
    model.train()
    return opt, model


def _net(model):
    n = model.net_ray_marching
    return n.module if hasattr(n, "module") else n


def test_reference_shell_trains_and_tests_on_the_device(ref_model):
    from pointnerf_amd import scenes
    from oracle import pyref
    opt, model = ref_model
    import models.mvs_points_volumetric_model as shell
    assert shell.__file__.startswith(U.REF) and type(model).__module__ == "models.mvs_points_volumetric_model"       # the reference's own shell
    assert type(_net(model)).__module__ == "pointnerf_amd.neural_points_volumetric_model"
    assert next(_net(model).parameters()).is_cuda and opt.vsize == [0.004, 0.004, 0.004] and opt.agg_dist_pers == 20
    d = scenes.block_rays(theta_deg=55.0, x0=398, y0=398, size=5)
    data = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
    data["id"] = torch.tensor([3])
    opt.ray_jitter = 0.0               # ours: pins the in-kernel jitter so that the step is comparable with the oracle
    sd = {k: v.detach().cpu().clone() for k, v in _net(model).state_dict().items()}
    mlp = {k[len("aggregator."):]: v for k, v in sd.items() if k.startswith("aggregator.")}
    pts = dict(xyz=sd["neural_points.xyz"], **{k: sd["neural_points." + k] for k in ("points_embeding", "points_conf", "points_dir", "points_color")})
    inp = pyref.to_torch_inputs(d)
    ref = pyref.render(opt, pts, mlp, inp)
    want = float(pyref.training_loss(opt, ref, inp))
    # ---- evaluation first (weights unchanged): run/train_ft.py:302-303  model.set_input(data); model.test()
    model.eval()
    model.set_input(data)
    out = model.test()
    col = out["coarse_raycolor"].detach().cpu()
    full = pyref.fill_invalid(ref, inp)["coarse_raycolor"]
    err = float((col - full).abs().max())
    print("reference shell .test() on libpnerf_hip.so: ray colour err %.2e over %d rays (%d hit)" % (err, col.shape[1], int(ref["ray_mask"].sum())))
    assert col.shape == full.shape and err <= 1e-4
    # ---- one optimisation step: run/train_ft.py:927,937
    model.train()
    model.set_input(data)
    model.optimize_parameters(total_steps=1)
    got = float(model.get_current_losses()["total"])
    print("loss through the reference's shell on the device kernels: %.8f, oracle: %.8f" % (got, want))
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want))
    assert not torch.equal(_net(model).state_dict()["aggregator.block1.0.weight"].cpu(), mlp["block1.0.weight"])       # both optimizers ran
    # the loaded native code is the product library, not the emulator
    maps = open("/proc/self/maps").read()
    assert "libpnerf_hip.so" in maps and "libpnerf_emu" not in maps
