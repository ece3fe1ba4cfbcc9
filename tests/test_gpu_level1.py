"""GPU test of the level-1 drop-ins used ONE BY ONE exactly as the reference's NeuralPointsRayMarching.forward chains
them (models/neural_points_volumetric_model.py:268-306): NeuralPoints.forward (14-tuple) -> PointAggregator.forward ->
ray_dist -> ray_march, forward and backward, against the oracle and against the fused path."""
import numpy as np
import pytest
import torch

from cases import CASES, build_case
from gpu_util import DEV
from pointnerf_amd import config, scenes
from pointnerf_amd.neural_points import NeuralPoints
from pointnerf_amd.point_aggregators import PointAggregator
from pointnerf_amd.neural_points_volumetric_model import NeuralPointsRayMarching
from pointnerf_amd.diff_ray_marching import ray_march, near_far_linear_ray_generation
from pointnerf_amd.diff_render_func import find_render_function, find_blend_function, find_tone_map
from oracle import pyref

pytestmark = pytest.mark.gpu


def _build(name):
    opt, xyz, attrs, inp, mlp = build_case(name)
    dev = torch.device(DEV)
    agg = PointAggregator(opt).to(dev)
    agg.load_state_dict(mlp)
    agg.flatten_()
    npnt = NeuralPoints(32, xyz.shape[0], opt, dev)
    a = {k: v.to(dev) for k, v in attrs.items()}
    npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"],
                    points_conf=a["points_conf"], parameter=True)
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    return opt, xyz, attrs, inp, mlp, agg, npnt, d


def _level1_forward(opt, agg, npnt, d):
    """The reference's forward body, verbatim in structure, on our modules."""
    sampled_color, sampled_Rw2c, sampled_dir, sampled_conf, sampled_embedding, sampled_xyz_pers, sampled_xyz, sample_pnt_mask, \
        sample_loc, sample_loc_w, sample_ray_dirs, ray_mask_tensor, vsize, grid_vox_sz = npnt(
            {"pixel_idx": d["pixel_idx"], "camrotc2w": d["camrotc2w"], "campos": d["campos"], "near": d["near"], "far": d["far"],
             "focal": None, "h": d["h"], "w": d["w"], "intrinsic": d["intrinsic"], "gt_image": d["gt_image"], "raydir": d["raydir"]})
    decoded_features, ray_valid, weight, conf_coefficient = agg(
        sampled_color, sampled_Rw2c, sampled_dir, sampled_conf, sampled_embedding, sampled_xyz_pers, sampled_xyz, sample_pnt_mask,
        sample_loc, sample_loc_w, sample_ray_dirs, vsize, grid_vox_sz)
    ray_dist = torch.cummax(sample_loc[..., 2], dim=-1)[0]
    ray_dist = torch.cat([ray_dist[..., 1:] - ray_dist[..., :-1],
                          torch.full((ray_dist.shape[0], ray_dist.shape[1], 1), vsize[2], device=ray_dist.device)], dim=-1)
    mask = ray_dist < 1e-8
    if opt.raydist_mode_unit > 0:
        mask = torch.logical_or(mask, ray_dist > 2 * vsize[2])
    mask = mask.to(torch.float32)
    ray_dist = ray_dist * (1.0 - mask) + mask * vsize[2]
    ray_dist *= ray_valid.float()
    ray_color, point_color, opacity, acc_transmission, blend_weight, background_transmission, bg_bw = ray_march(
        ray_dist, ray_valid, decoded_features, find_render_function("radiance"), find_blend_function("alpha"), d["bg_color"])
    ray_color = find_tone_map("off")(ray_color)
    return dict(coarse_raycolor=ray_color, coarse_point_opacity=opacity, coarse_is_background=background_transmission,
                ray_mask=ray_mask_tensor, weight=weight, blend_weight=blend_weight, conf_coefficient=conf_coefficient,
                decoded_features=decoded_features, acc_transmission=acc_transmission, point_color=point_color, bg_bw=bg_bw)


@pytest.mark.parametrize("name", list(CASES))
def test_level1_chain_matches_oracle_and_fused(name):
    opt, xyz, attrs, inp, mlp, agg, npnt, d = _build(name)
    out = _level1_forward(opt, agg, npnt, d)
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    op = dict(xyz=xyz, **{k: v.clone().requires_grad_(True) for k, v in attrs.items()})
    ref = pyref.render(opt, op, om, inp)
    assert torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])
    for k, rk in [("coarse_raycolor", "coarse_raycolor"), ("coarse_point_opacity", "coarse_point_opacity"),
                  ("coarse_is_background", "coarse_is_background"), ("decoded_features", "decoded_features"),
                  ("weight", "weight"), ("blend_weight", "blend_weight"), ("conf_coefficient", "conf_coefficient")]:
        a, b = out[k].detach().cpu(), ref[rk].detach()
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert float((a - b).abs().max()) <= 1e-4, k
    # acc_transmission: exclusive product, first entry 1
    assert float((out["acc_transmission"][..., 0] - 1).abs().max()) == 0.0
    assert torch.equal(out["bg_bw"], out["coarse_is_background"])
    # gradients through the level-1 chain
    loss = pyref.training_loss(opt, out, {"gt_image": d["gt_image"]})
    loss.backward()
    pyref.training_loss(opt, ref, inp).backward()
    for n, p in agg.named_parameters():
        g, r = p.grad.cpu(), om[n].grad
        assert float((g - r).abs().max()) <= 2e-3 * max(float(r.abs().max()), 1e-8), n
    for n in ("points_embeding", "points_conf", "points_color", "points_dir"):
        g, r = getattr(npnt, n).grad.cpu(), op[n].grad
        e = (g - r).abs()
        assert float((e > 1e-3 * float(r.abs().max())).float().mean()) <= 1e-3, n
    # fused path gives the same numbers
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    with torch.no_grad():
        fo = model(**d)
    assert float((fo["coarse_raycolor"] - out["coarse_raycolor"].detach()).abs().max()) <= 2e-6
    assert torch.equal(fo["ray_mask"], out["ray_mask"])


def test_ray_generation_dropin_is_bit_exact_on_device():
    fix = np.load(__file__.replace("test_gpu_level1.py", "golden/raygen_pe.npz"))
    inp = pyref.to_torch_inputs(scenes.block_rays(size=4))
    raypos, seg, valid, mid = near_far_linear_ray_generation(inp["campos"].to(DEV), inp["raydir"].to(DEV), 400, near=2.0, far=6.0, jitter=0.0)
    assert raypos.shape == (1, 16, 400, 3)
    # the device cumsum may round differently from the CPU one: allow 1 ulp of the depth range
    assert np.abs(raypos.cpu().numpy() - fix["raypos"]).max() <= 2e-6


def test_ray_march_rejects_other_funcs():
    from pointnerf_amd.diff_render_func import white_color, alpha_blend
    x = torch.zeros(1, 2, 4, device=DEV)
    with pytest.raises(NotImplementedError):
        ray_march(x, x > 0, torch.zeros(1, 2, 4, 4, device=DEV), white_color, alpha_blend)


def test_probe_outputs_prob1():
    """opt.prob == 1 (probe_hole, run/train_ft.py:417-530): the 7 extra outputs against the oracle."""
    opt, xyz, attrs, inp, mlp, agg, npnt, d = _build("small_k8")
    opt.prob = 1
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    with torch.no_grad():
        out = model(**d)
        points = dict(xyz=xyz, **attrs)
        ref = pyref.render(opt, points, mlp, inp)
        pr = pyref.probe_outputs(ref, points)
    for k, v in pr.items():
        a = out[k].cpu()
        assert a.shape == v.shape, (k, a.shape, v.shape)
        assert float((a - v).abs().max()) <= 1e-4, k
    opt.prob = 0


def test_normview_rotation_rw2c_and_no_background():
    """NeuralPoints.Rw2c != identity (scenes initialised with normview, point_aggregators.py:492-496,506,526,566) and
    bg_color=None ('bg_ray' callers): forward + gradients against the oracle."""
    opt, xyz, attrs, inp, mlp, agg, npnt, d = _build("small_k4")
    g = torch.Generator().manual_seed(5)
    qm, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    npnt.Rw2c = qm.to(DEV)
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    d2 = dict(d); d2["bg_ray"] = torch.zeros(1, d["raydir"].shape[1], 3, device=DEV)      # forward() then ignores bg_color
    out = model(**d2)
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    op = dict(xyz=xyz, **{k: v.clone().requires_grad_(True) for k, v in attrs.items()})
    inp2 = dict(inp); inp2["bg_color"] = None
    ref = pyref.render(opt, op, om, inp2, Rw2c=qm)
    assert float((out["coarse_raycolor"].cpu() - ref["coarse_raycolor"]).abs().max()) <= 1e-4
    (out["coarse_raycolor"].sum() * 1.0).backward()
    ref["coarse_raycolor"].sum().backward()
    for n, p in agg.named_parameters():
        gh, r = p.grad.cpu(), om[n].grad
        assert float((gh - r).abs().max()) <= 2e-3 * max(float(r.abs().max()), 1e-8), n
    for n in ("points_dir", "points_color", "points_conf"):
        gh, r = getattr(npnt, n).grad.cpu(), op[n].grad
        assert float((gh - r).abs().max()) <= 2e-3 * max(float(r.abs().max()), 1e-8), n


def test_full_image_eval_loop_matches_oracle():
    """pointnerf_amd.eval_loop.render_image (chunked, canvas on the device) == the oracle rendered ray by ray."""
    from pointnerf_amd import eval_loop
    opt, xyz, attrs, inp, mlp, agg, npnt, d = _build("small_k8")
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    # a 24x20 window of the 800x800 camera: shift the principal point so that window pixel (0,0) is image pixel (388,390)
    intr = inp["intrinsic"][0].clone()
    intr[0, 2] -= 388.0; intr[1, 2] -= 390.0
    h, w = 20, 24
    img, hit = eval_loop.render_image(model, d["campos"], d["camrotc2w"], intr, h, w, d["near"], d["far"], d["bg_color"], chunk=157)
    pix = eval_loop.pixel_grid(h, w, torch.device("cpu"))
    sub = dict(inp)
    sub["raydir"] = eval_loop.rays_from_pixels(pix, intr, inp["camrotc2w"])
    with torch.no_grad():
        ref = pyref.render(opt, dict(xyz=xyz, **attrs), mlp, sub)
        full = pyref.fill_invalid(ref, sub)
    assert torch.equal(hit.cpu(), ref["ray_mask"][0] > 0)
    assert float((img.cpu().reshape(-1, 3) - full["coarse_raycolor"][0]).abs().max()) <= 1e-4
    gt = torch.rand(h, w, 3, generator=torch.Generator().manual_seed(0))
    p1 = float(eval_loop.psnr(img, gt)); p2 = float(-10 * torch.log10(torch.mean((full["coarse_raycolor"][0] - gt.reshape(-1, 3)) ** 2)))
    assert abs(p1 - p2) < 1e-3


def test_eval_loop_chunk_that_hits_nothing():
    """A chunk of an image may see no geometry at all (sky): the model must return the background for it, not fail on the
    empty gathers (found with a full 800 x 800 view)."""
    from pointnerf_amd import eval_loop
    opt, xyz, attrs, inp, mlp, agg, npnt, d = _build("small_k8")
    opt.prob = 1                                      # the probe outputs gather per-point tensors for the hit rays: empty here
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    intr = inp["intrinsic"][0].clone()
    intr[0, 2] += 5000.0                              # look far off to the side: no ray meets the cloud
    img, hit = eval_loop.render_image(model, d["campos"], d["camrotc2w"], intr, 8, 8, d["near"], d["far"], d["bg_color"], chunk=32)
    assert not bool(hit.any())
    assert float((img.reshape(-1, 3) - d["bg_color"].reshape(1, 3)).abs().max()) <= 1e-6
    opt.prob = 0


def test_standalone_aggregator_with_holes_in_the_neighbor_mask():
    """PointAggregator.forward is a public entry of its own: a caller may pass any sample_pnt_mask, not only the
    front-filled slots the query produces.  The sample classes of the HIP path (rows per sample = K, K/2 or K/4) must then be
    chosen by the LAST occupied slot: knock random slots out of a real query result and compare forward + gradients with the
    oracle on the same punched mask."""
    opt, xyz, attrs, inp, mlp, agg, npnt, d = _build("small_k8")
    t = npnt({"pixel_idx": d["pixel_idx"], "camrotc2w": d["camrotc2w"], "campos": d["campos"], "near": d["near"], "far": d["far"],
              "focal": None, "h": d["h"], "w": d["w"], "intrinsic": d["intrinsic"], "gt_image": d["gt_image"], "raydir": d["raydir"]})
    sampled_color, sampled_Rw2c, sampled_dir, sampled_conf, sampled_embedding, sampled_xyz_pers, sampled_xyz, mask, \
        sample_loc, sample_loc_w, sample_ray_dirs, ray_mask_tensor, vsize, grid_vox_sz = t
    gen = torch.Generator().manual_seed(11)
    keep = (torch.rand(mask.shape, generator=gen) > 0.45).to(mask.device)
    punched = mask & keep                                        # holes anywhere: e.g. slots {0, 5} of 8 occupied
    assert bool((punched[..., 0] != punched[..., 1]).any())
    out, ray_valid, weight, conf_c = agg(sampled_color, sampled_Rw2c, sampled_dir, sampled_conf, sampled_embedding, sampled_xyz_pers,
                                         sampled_xyz, punched, sample_loc, sample_loc_w, sample_ray_dirs, vsize, grid_vox_sz)
    # oracle on the same gathered tensors and the same mask
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    nb = dict(mask=punched.cpu(), xyz=sampled_xyz.detach().cpu(), xyz_pers=sampled_xyz_pers.detach().cpu(), emb=sampled_embedding.detach().cpu(),
              color=sampled_color.detach().cpu(), dir=sampled_dir.detach().cpu(), conf=sampled_conf.detach().cpu())
    ro, rv, rw, rc = pyref.aggregate(opt, om, nb, sample_loc.detach().cpu(), sample_loc_w.detach().cpu(), sample_ray_dirs.detach().cpu())
    assert torch.equal(ray_valid.cpu(), rv)
    assert float((out.detach().cpu() - ro.detach()).abs().max()) <= 1e-4
    assert float((weight.detach().cpu() - rw.detach()).abs().max()) <= 1e-5
    gen2 = torch.Generator().manual_seed(12)
    probe = torch.randn(ro.shape, generator=gen2)
    (out * probe.to(out.device)).sum().backward()
    (ro * probe).sum().backward()
    for n, p in agg.named_parameters():
        g, r = p.grad.cpu(), om[n].grad
        assert float((g - r).abs().max()) <= 2e-3 * max(float(r.abs().max()), 1e-8), n


def test_prune_and_grow_between_steps_keep_parity():
    """The probe-and-grow / prune step of the training loop (run/train_ft.py:417-530, neural_points.py:347-399) changes the
    point cloud between two renders: the cached voxel grid must be rebuilt and the render of the new cloud must match the
    oracle on the new cloud."""
    opt, xyz, attrs, inp, mlp, agg, npnt, d = _build("small_k8")
    model = NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
    with torch.no_grad():
        model(**d)                                            # builds and caches the grid of the full cloud
    thresh = 0.55
    keep = attrs["points_conf"][0, :, 0] >= thresh
    npnt.prune(thresh)
    assert npnt.xyz.shape[0] == int(keep.sum())
    gen = torch.Generator().manual_seed(5)
    add = 300
    new_xyz = xyz[keep][:add] + 0.002 * torch.randn(add, 3, generator=gen)
    new_emb = torch.rand(add, 32, generator=gen) - 0.5
    new_col, new_dir, new_conf = torch.rand(add, 3, generator=gen), torch.nn.functional.normalize(torch.randn(add, 3, generator=gen), dim=-1), 0.2 + 0.7 * torch.rand(add, 1, generator=gen)
    dev = npnt.xyz.device
    npnt.grow_points(new_xyz.to(dev), new_emb.to(dev), new_col.to(dev), new_dir.to(dev), new_conf.to(dev))
    with torch.no_grad():
        out = model(**d)
    xyz2 = torch.cat([xyz[keep], new_xyz], 0)
    attrs2 = dict(points_embeding=torch.cat([attrs["points_embeding"][:, keep], new_emb[None]], 1),
                  points_conf=torch.cat([attrs["points_conf"][:, keep], new_conf[None]], 1),
                  points_dir=torch.cat([attrs["points_dir"][:, keep], new_dir[None]], 1),
                  points_color=torch.cat([attrs["points_color"][:, keep], new_col[None]], 1))
    with torch.no_grad():
        ref = pyref.render(opt, dict(xyz=xyz2, **attrs2), mlp, inp)
        full = pyref.fill_invalid(ref, inp)
    assert torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])
    assert float((out["coarse_raycolor"].cpu() - full["coarse_raycolor"]).abs().max()) <= 1e-4


def test_fused_zero_one_loss_matches_the_torch_chain():
    """ops.ZeroOneConf against gather + gradient_clamp + clamp + logs + sum on the device, 20 M slots with the point-0 flood"""
    from pointnerf_amd import ops
    from pointnerf_amd.neural_points_volumetric_model import gradient_clamp
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    N, M, eps = 300_000, 20_000_000, 1e-3
    conf = (torch.rand(1, N, 1, generator=g) * 1.2 - 0.1).to(dev)
    pidx = torch.randint(-1, N, (M,), generator=g, dtype=torch.int32)
    pidx[torch.rand(M, generator=g) < 0.5] = -1
    pidx = pidx.to(dev)
    a = conf.clone().requires_grad_(True)
    cc = gradient_clamp(ops.gather_rows(a.reshape(-1, 1), pidx)[..., 0])
    v = cc.clamp(eps, 1 - eps)
    ref = (torch.log(v) + torch.log(1 - v)).sum() / M
    ref.backward()
    b = conf.clone().requires_grad_(True)
    got = ops.zero_one_conf_sum(b, pidx, eps) / M
    got.backward()
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref))
    scale = float(a.grad.abs().max())
    assert float((b.grad - a.grad).abs().max()) <= 1e-4 * scale      # point 0 collects ~1e7 terms on both sides: fp32 summation order
