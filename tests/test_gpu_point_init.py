"""pnerf_voxel_downsample (csrc/pointinit.hip) through the drop-in ``construct_vox_points_closest`` against the oracle:
voxel list and closest-member indices bit-exact, centroids to fp32 rounding; then a lego-sized cloud with properties."""
import pytest
import torch

from pointnerf_amd import point_init, scenes
from oracle import pyref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,res,box", [(5000, 24, None), (20000, 40, None), (3000, 7, ([-0.5, -0.5, -0.5], [0.5, 0.7, 0.9])), (1, 5, None)])
def test_voxel_downsample_matches_oracle(n, res, box):
    gen = torch.Generator().manual_seed(n)
    xyz = torch.randn(n, 3, generator=gen) * torch.tensor([0.4, 0.3, 0.2]) + 0.05
    if n == 1:
        xyz = torch.tensor([[0.1, 0.2, 0.3]])
    kw = {} if box is None else dict(space_min=box[0], space_max=box[1])
    if n == 1:
        kw = dict(space_min=[0.0, 0.0, 0.0], space_max=[1.0, 1.0, 1.0])     # a single point has no extent of its own
    cen_o, gidx_o, midx_o, outside = pyref.vox_points_closest(xyz, res, **kw)
    cen, gidx, midx = point_init.construct_vox_points_closest(xyz.cuda(), res, **kw)
    assert torch.equal(gidx.cpu(), gidx_o)
    assert torch.equal(midx.cpu(), midx_o)
    assert float((cen.cpu() - cen_o).abs().max()) <= 1e-6
    assert torch.equal(point_init.construct_vox_points_xyz(xyz.cuda(), res, **kw).cpu(), cen.cpu())


def test_voxel_downsample_lego_scale_properties():
    """2 M raw points at the lego script's vox_res = 320 (lego_cuda.sh:17): every chosen index lies in its voxel, voxels are
    strictly ascending, every input point is accounted for, and two runs agree bit for bit."""
    xyz = torch.from_numpy(scenes.lego_points(2_000_000)).cuda()
    cen, gidx, midx = point_init.construct_vox_points_closest(xyz, 320)
    cen2, gidx2, midx2 = point_init.construct_vox_points_closest(xyz, 320)
    assert torch.equal(cen, cen2) and torch.equal(gidx, gidx2) and torch.equal(midx, midx2)
    key = (gidx[:, 0].long() * 320 + gidx[:, 1].long()) * 320 + gidx[:, 2].long()
    assert bool((key[1:] > key[:-1]).all())
    smin, vox = point_init._space(xyz, 320, None, None)
    cell = torch.floor((xyz - smin.cuda()[None]) / vox.cuda()[None]).to(torch.int32)
    assert torch.equal(cell[midx], gidx)
    assert len(torch.unique(cell, dim=0)) == len(gidx)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_extract_2d_and_query_embedding(tag):
    """MvsPointsModel.query_embedding on the device (csrc/embed2d.hip) = the oracle's restatement = the reference's own extract_2d /
    query_embedding / homo_warp_nongrid(_occ) / extract_from_2d_grid exec'ed from source (tests/golden/refembed.npz): both mask variants,
    one and three source views, the current camera among them or not, colours, per-view directions."""
    from embed_case import run_case
    err = run_case(tag, "cuda:0")
    print(tag, err)
    assert max(err.values()) <= 1e-5, err


@pytest.mark.gpu
def test_extract_2d_at_scale_properties():
    """2 M points x 3 views x the full pyramid: size-independent properties -- rows outside every mask are zero, rows inside are a convex
    combination of the map's values (min <= value <= max per channel), the colour columns of a constant image are that constant, and the call
    is deterministic."""
    import types
    from shell_fakes import embed_inputs
    from pointnerf_amd.mvs_points_model import MvsPointsModel
    inp = embed_inputs(seed=9, n=2_000_000, HD=96, WD=128, focal=104.0)
    dev = "cuda:0"
    for occ in (0, 1):
        m = MvsPointsModel(types.SimpleNamespace(depth_occ=occ, ref_vid=0, shading_feature_mlp_layer0=0))
        feats = [f.to(dev) for f in inp["img_feats"]]
        feats[0] = torch.full_like(feats[0], 0.25)
        args = (feats, [0, 1, 2], [0, 1, 2, 3], inp["intrinsics"].to(dev), inp["c2ws"].to(dev), inp["w2cs"].to(dev), inp["cam_xyz"].to(dev),
                inp["HD"], inp["WD"])
        f1, c1, mask = m.extract_2d(*args, cam_vid=0, return_mask=True)
        f2, c2 = m.extract_2d(*args, cam_vid=0)
        assert torch.equal(f1, f2) and torch.equal(c1, c2)
        assert f1.shape == (1, 2_000_000, 168) and c1.shape == (1, 2_000_000, 9)
        for v in range(3):
            mv = mask[v]
            assert (0.02 if occ == 0 else 0.001) < float(mv.float().mean()) < 0.98      # (z-buffer of 96 x 128 pixels: ~1 % of 2 M points survive)
            fv, cv = f1[0, :, 56 * v:56 * (v + 1)], c1[0, :, 3 * v:3 * (v + 1)]
            assert float(fv[~mv].abs().max()) == 0.0 and float(cv[~mv].abs().max()) == 0.0
            assert float((cv[mv] - 0.25).abs().max()) <= 1e-6
            col = 0
            for lid in (1, 2, 3):
                fm = feats[lid][v]
                lo, hi = fm.amin(dim=(1, 2)), fm.amax(dim=(1, 2))
                blk = fv[mv][:, col:col + fm.shape[0]]
                assert bool((blk >= lo - 1e-6).all()) and bool((blk <= hi + 1e-6).all())
                col += fm.shape[0]
