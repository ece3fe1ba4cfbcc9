"""CPU test: the checkpoint wire format.  The reference saves ``net_ray_marching.module.state_dict()``
(models/base_model.py:85-102) whose keys are ``neural_points.{xyz,points_embeding,points_conf,points_dir,points_color}``
(models/neural_points/neural_points.py:243-288) and ``aggregator.<seq>.<i>.{weight,bias}``; published checkpoints must
load into our modules unchanged, and ours into the reference."""
import torch

from pointnerf_amd import config, scenes
from pointnerf_amd.neural_points import NeuralPoints
from pointnerf_amd.point_aggregators import PointAggregator
from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching

REF_KEYS = ["neural_points.xyz", "neural_points.points_embeding", "neural_points.points_conf", "neural_points.points_dir",
            "neural_points.points_color",
            "aggregator.block1.0.weight", "aggregator.block1.0.bias", "aggregator.block1.2.weight", "aggregator.block1.2.bias",
            "aggregator.block3.0.weight", "aggregator.block3.0.bias", "aggregator.block3.2.weight", "aggregator.block3.2.bias",
            "aggregator.alpha_branch.0.weight", "aggregator.alpha_branch.0.bias",
            "aggregator.color_branch.0.weight", "aggregator.color_branch.0.bias", "aggregator.color_branch.2.weight",
            "aggregator.color_branch.2.bias", "aggregator.color_branch.4.weight", "aggregator.color_branch.4.bias",
            "aggregator.color_branch.6.weight", "aggregator.color_branch.6.bias"]


def _model(n=64, seed=0):
    opt = config.lego_opt()
    dev = torch.device("cpu")
    agg = PointAggregator(opt)
    npnt = NeuralPoints(32, n, opt, dev)
    a = {k: torch.from_numpy(v) for k, v in scenes.point_attributes(n, 32, seed).items()}
    npnt.set_points(torch.from_numpy(scenes.chair_points(n, seed=seed)), a["points_embeding"], points_color=a["points_color"],
                    points_dir=a["points_dir"], points_conf=a["points_conf"], parameter=True)
    return opt, NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)


def test_state_dict_keys_and_shapes_are_the_reference_ones():
    opt, m = _model()
    sd = m.state_dict()
    assert sorted(sd.keys()) == sorted(REF_KEYS)
    assert sd["neural_points.xyz"].shape == (64, 3) and sd["neural_points.points_embeding"].shape == (1, 64, 32)
    assert sd["neural_points.points_conf"].shape == (1, 64, 1) and sd["neural_points.points_dir"].shape == (1, 64, 3)
    assert sd["aggregator.block1.0.weight"].shape == (256, 284) and sd["aggregator.block3.0.weight"].shape == (256, 263)
    assert sd["aggregator.color_branch.0.weight"].shape == (128, 280) and sd["aggregator.alpha_branch.0.weight"].shape == (1, 256)
    # requires_grad follows the script's flags (lego_cuda.sh:12-15; xyz_grad default 0)
    p = dict(m.named_parameters())
    assert not p["neural_points.xyz"].requires_grad and p["neural_points.points_embeding"].requires_grad
    # optimizer split of the reference: names containing "neural_points" go to the point optimizer (neural_points_volumetric_model.py:189-190)
    assert sum(v.numel() for k, v in p.items() if "neural_points" not in k) == 341764


def test_checkpoint_round_trip_through_the_reference_loader_path(tmp_path):
    opt, m = _model(seed=1)
    path = str(tmp_path / "100_net_ray_marching.pth")
    torch.save({k: v.cpu() for k, v in m.state_dict().items()}, path)            # base_model.py:90-98
    # NeuralPoints(checkpoint=path) is how the reference restores the cloud (neural_points.py:240-288)
    opt2 = config.lego_opt(load_points=1)
    npnt = NeuralPoints(32, 64, opt2, torch.device("cpu"), checkpoint=path)
    for k in ("xyz", "points_embeding", "points_conf", "points_dir", "points_color"):
        assert torch.equal(getattr(npnt, k).data, m.state_dict()["neural_points." + k]), k
    agg = PointAggregator(opt2)
    saved = torch.load(path)
    missing = agg.load_state_dict({k[len("aggregator."):]: v for k, v in saved.items() if k.startswith("aggregator.")}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


def test_prune_grow_set_points_semantics():
    """neural_points.py:347-399: prune keeps conf >= thresh, grow appends, parameters are re-created (which is what
    invalidates the cached voxel grid: the cache is keyed on the storage)."""
    opt, m = _model(n=100, seed=2)
    npnt = m.neural_points
    conf = npnt.points_conf.data
    keep = int((conf[0, :, 0] >= 0.5).sum())
    old_ptr = npnt.xyz.data_ptr()
    npnt.prune(0.5)
    assert npnt.xyz.shape == (keep, 3) and npnt.points_embeding.shape == (1, keep, 32) and npnt.points_conf.shape == (1, keep, 1)
    assert bool((npnt.points_conf.data >= 0.5).all()) and npnt.xyz.data_ptr() != old_ptr
    add = 7
    npnt.grow_points(torch.zeros(add, 3), torch.zeros(add, 32), torch.zeros(add, 3), torch.zeros(add, 3), torch.ones(add, 1))
    assert npnt.xyz.shape == (keep + add, 3) and npnt.points_color.shape == (1, keep + add, 3)
    assert isinstance(npnt.points_dir, torch.nn.Parameter) and npnt.points_dir.requires_grad
    names = [k for k, _ in npnt.named_parameters()]
    assert sorted(names) == ["points_color", "points_conf", "points_dir", "points_embeding", "xyz"]
