"""Oracle of the point initialisation (voxel down-sampling) against what of the reference runs here
(torch.unique(dim=0) of the voxel coordinates, mvs_utils.py:553) and against brute-force properties."""
import torch

from oracle import pyref


def test_vox_oracle_voxel_list_is_torch_unique_and_members_are_consistent():
    gen = torch.Generator().manual_seed(3)
    xyz = torch.rand(5000, 3, generator=gen) * torch.tensor([1.0, 0.6, 0.3]) + torch.tensor([-0.2, 0.1, 2.0])
    res = 24
    cen, gidx, midx, outside = pyref.vox_points_closest(xyz, res)
    assert outside == 0
    # the reference's own statements for the voxel list (mvs_utils.py:541-553)
    xyz_min, xyz_max = xyz.min(0)[0], xyz.max(0)[0]
    space_edge = torch.max(xyz_max - xyz_min) * 1.05
    space_min = (xyz_max + xyz_min) / 2 - space_edge / 2
    vs = space_edge / res
    coords = torch.floor((xyz - space_min[None]) / vs).to(torch.int32)
    uniq, inv = torch.unique(coords, dim=0, return_inverse=True)
    assert torch.equal(uniq, gidx)
    for v in range(0, len(uniq), 37):
        mem = torch.nonzero(inv == v).reshape(-1)
        assert int(midx[v]) in mem.tolist()
        assert float((xyz[mem].mean(0) - cen[v]).abs().max()) <= 1e-6
        r = (xyz[mem] - cen[v]).norm(dim=-1)
        assert float(r.min()) >= float((xyz[int(midx[v])] - cen[v]).norm()) - 1e-7


def test_vox_oracle_box_drops_outside_points():
    gen = torch.Generator().manual_seed(4)
    xyz = torch.randn(2000, 3, generator=gen)
    cen, gidx, midx, outside = pyref.vox_points_closest(xyz, 10, space_min=[-1.0, -1.0, -1.0], space_max=[1.0, 1.0, 1.0])
    inside = ((xyz >= -1.0) & (xyz < 1.0)).all(-1)
    assert outside == int((~inside).sum())
    assert bool(inside[midx].all()) and int(gidx.min()) >= 0 and int(gidx.max()) < 10
