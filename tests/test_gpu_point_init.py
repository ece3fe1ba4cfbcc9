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
