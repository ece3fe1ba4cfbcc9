"""One training step of the REFERENCE's model shell on the overlay, in a fresh interpreter (tests/test_reference_overlay.py runs it with
PNERF_OVERLAY_FUSED=0: then NeuralPointsRayMarching.forward is the reference's own body, run module by module on the overlay's
NeuralPoints / PointAggregator / ray_march).  Prints one JSON line {"loss": ..., "oracle": ..., "network": module of the network class}."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_overlay_util as U  # noqa: E402


def main():
    opt = U.parse_options(["--gpu_ids", "-1", "--num_point", "1200", "--checkpoints_dir", "/tmp/pnerf_overlay_ckpt", "--resume_dir", "/tmp/pnerf_overlay_none",
                           "--SR", "12", "--K", "8", "--P", "24", "--max_o", "50000", "--ranges", "-0.3", "-0.3", "-0.3", "0.3", "0.3", "0.3",
                           "--random_sample_size", "5"])
    opt.mode = 2
    opt.is_train = True
    from emu_util import emu_backend
    from pointnerf_amd import scenes
    from oracle import pyref, query as oq
    oq.build()
    from models import create_model
    with emu_backend():
        model = create_model(opt)
        model.net_ray_marching = torch.nn.DataParallel(model.net_ray_marching)
        n = 1200
        xyz = torch.from_numpy(scenes.chair_points(n, seed=5, radius=0.06))
        a = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, 32, 5).items()}
        model.set_points(xyz, a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"])
        model.setup(opt, train_len=100)
        model.train()
        d = scenes.block_rays(theta_deg=55.0, x0=398, y0=398, size=5)
        data = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        data["id"] = torch.tensor([3])
        opt.ray_jitter = 0.0
        sd = {k: v.detach().clone() for k, v in model.net_ray_marching.module.state_dict().items()}
        model.set_input(data)
        model.optimize_parameters(total_steps=1)
        loss = float(model.get_current_losses()["total"])
    mlp = {k[len("aggregator."):]: v for k, v in sd.items() if k.startswith("aggregator.")}
    pts = dict(xyz=sd["neural_points.xyz"], **{k: sd["neural_points." + k] for k in ("points_embeding", "points_conf", "points_dir", "points_color")})
    inp = pyref.to_torch_inputs(d)
    want = float(pyref.training_loss(opt, pyref.render(opt, pts, mlp, inp), inp))
    net = type(model.net_ray_marching.module)
    changed = not torch.equal(model.net_ray_marching.module.state_dict()["aggregator.block1.0.weight"], mlp["block1.0.weight"])
    print(json.dumps({"loss": loss, "oracle": want, "network": net.__module__, "updated": changed}))


if __name__ == "__main__":
    main()
