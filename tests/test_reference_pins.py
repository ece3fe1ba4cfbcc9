"""Pins of the four blocks of the reference that cannot be imported here: the oracle's
restatements AND the product's host code against tests/golden/refshell.npz, which tests/golden/make_golden.py --shell wrote by
exec'ing the reference's own source text (models/neural_points_volumetric_model.py:331-362, models/mvs/mvs_utils.py:537-561,
run/train_ft.py:252-414 and :417-540) on the seeded stand-ins of tests/shell_fakes.py.  Runs wherever the fixture is (no
/root/reference needed)."""
import os

import numpy as np
import pytest
import torch

import shell_fakes as SF
from cases import build_case
from oracle import pyref
from pointnerf_amd import eval_loop, probe

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refshell.npz"))


def test_probe_outputs_of_the_oracle_equal_the_reference_forward_body():
    """opt.prob == 1 outputs (neural_points_volumetric_model.py:331-362): same tensors in, bit-identical tensors out"""
    opt, xyz, attrs, inp, mlp = build_case("small_k8")
    points = dict(xyz=xyz, **attrs)
    out = pyref.render(opt, points, mlp, inp)
    got = pyref.probe_outputs({k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}, points)
    for k in ("ray_max_shading_opacity", "ray_max_sample_loc_w", "ray_max_far_dist", "shading_avg_color", "shading_avg_dir",
              "shading_avg_conf", "shading_avg_embedding"):
        ref = GOLD["prob_" + k]
        assert got[k].shape == ref.shape, k
        assert np.array_equal(got[k].numpy(), ref), (k, float(np.abs(got[k].numpy() - ref).max()))


@pytest.mark.parametrize("tag,n,res,seed,box", [("a", 5000, 24, 3, False), ("b", 20000, 40, 5, False), ("c", 3000, 16, 7, True)])
def test_vox_downsampling_of_the_oracle_equals_construct_vox_points_closest(tag, n, res, seed, box):
    """mvs_utils.py:537-561 exec'ed with pure-torch scatter_mean / scatter_min: voxel list and closest member identical,
    centroids to the last bit (both sum in point order)"""
    pts = SF.vox_cloud(n, seed)
    kw = dict(space_min=pts.min(0)[0] - 0.01, space_max=pts.max(0)[0] + 0.01) if box else {}
    cen, gidx, midx, outside = pyref.vox_points_closest(pts, res, **kw)
    assert outside == 0
    assert np.array_equal(gidx.numpy(), GOLD["vox_%s_grid" % tag])
    assert np.array_equal(cen.numpy(), GOLD["vox_%s_centroid" % tag])
    ref_idx = GOLD["vox_%s_min_idx" % tag]
    diff = np.nonzero(midx.numpy() != ref_idx)[0]
    # torch.norm and the oracle's sqrt((dx^2 + dy^2) + dz^2) may round a residual differently: a different member is only
    # acceptable where the two candidates are equally close to one ulp
    for v in diff:
        c = cen[v]
        r0, r1 = float((pts[int(midx[v])] - c).norm()), float((pts[int(ref_idx[v])] - c).norm())
        assert abs(r0 - r1) <= 2e-7 * max(r0, r1), (v, r0, r1)
    assert len(diff) <= 2, len(diff)


@pytest.mark.parametrize("tag,far_thresh", [("near", -1.0), ("far", 0.012)])
def test_probe_hole_of_the_product_equals_the_reference_function(tag, far_thresh):
    """run/train_ft.py:417-540 exec'ed on the stand-ins: the same candidates in the same order, values identical (incl. the
    repeated prob_mul scaling of add_conf); the product renders a view in larger chunks, which must not matter"""
    o, model, data = SF.shell_probe_setup(far_thresh)
    H, W = data.height, data.width
    xyz, emb, color, dirs, conf = probe.probe_hole(model, data, o, H, W, test_steps=150, opacity_thresh=0.3, chunk=173)
    assert set(model.seen) == {(1, (7, 7, 7))} and getattr(model.opt, "prob", 0) == 0 and list(model.opt.query_size) == [3, 3, 3]
    for name, t in (("xyz", xyz), ("embedding", emb), ("color", color), ("dir", dirs), ("conf", conf)):
        ref = GOLD["probe_%s_%s" % (tag, name)]
        assert tuple(t.shape) == ref.shape, (name, t.shape, ref.shape)
        assert np.array_equal(t.numpy(), ref), name
    # the oracle's index-loop mask selects the same pixels, view by view (frame order of the reference: ranking 2, 1, 0)
    rows = []
    for i in (2, 1, 0):
        hit, maps = model.maps(i)
        v = data[i]
        pl = v["pixel_idx"].reshape(-1, 2).long()
        edge = np.zeros((H, W), bool); edge[pl[:, 1].numpy(), pl[:, 0].numpy()] = True
        hm = (hit & torch.from_numpy(edge)).numpy()
        msk = lambda t: (t * torch.from_numpy(hm)[..., None]).numpy()
        col = np.where(hm[..., None], maps["coarse_raycolor"].numpy(), 1.0).astype(np.float32)
        col[~edge] = 0
        m = pyref.probe_hole_mask(hm.astype(np.float32), msk(maps["ray_max_shading_opacity"])[..., 0], msk(maps["ray_max_far_dist"])[..., 0],
                                  col, data.gt_canvas(i).reshape(H, W, 3).numpy(), np.ones((1, 3), np.float32), edge, 0.3, far_thresh)
        rows.append(maps["ray_max_sample_loc_w"].numpy()[m])
    assert np.array_equal(np.concatenate(rows), GOLD["probe_%s_xyz" % tag])


def test_eval_loop_of_the_product_equals_the_reference_test_function():
    """run/train_ft.py:252-414 exec'ed on the stand-ins: per-view losses of both items, the PSNR it returns, the canvases"""
    o, model, data = SF.shell_test_setup()
    H, W = data.height, data.width
    seen = []
    psnr, avg = eval_loop.test_views(model, data, o, H, W, test_num_step=o.test_num_step, chunk=100, on_view=lambda i, vis: seen.append((i, vis)))
    ref_items = GOLD["test_items"]                      # [views, (coarse_raycolor, ray_masked_coarse_raycolor)]
    assert [i for i, _ in seen] == [0, 2, 4] and ref_items.shape == (3, 2)
    assert abs(psnr - float(GOLD["test_psnr"])) <= 1e-5 * abs(float(GOLD["test_psnr"]))
    assert abs(avg["coarse_raycolor"] - ref_items[:, 0].mean()) <= 1e-6 * ref_items[:, 0].mean()
    assert abs(avg["ray_masked_coarse_raycolor"] - ref_items[:, 1].mean()) <= 1e-6 * ref_items[:, 1].mean()
    for j, (i, vis) in enumerate(seen):
        assert np.array_equal(vis["coarse_raycolor"].numpy(), GOLD["test_canvas"][j].astype(np.float32)), i
        # the oracle's restatement of the per-view block
        v = data[i]
        hit, _ = model.maps(i)
        pl = v["pixel_idx"].reshape(-1, 2).long().numpy()
        r = pyref.test_view_losses(vis["coarse_raycolor"].numpy(), v["gt_image"][0].numpy(), pl, hit.numpy()[pl[:, 1], pl[:, 0]], H, W)
        assert abs(r["coarse_raycolor"] - ref_items[j, 0]) <= 2e-6 * ref_items[j, 0]
        assert abs(r["ray_masked_coarse_raycolor"] - ref_items[j, 1]) <= 2e-6 * ref_items[j, 1]
