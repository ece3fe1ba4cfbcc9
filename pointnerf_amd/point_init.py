"""Point initialisation that feeds the hot path: voxel down-sampling of a raw point cloud into neural-point
positions.  Drop-in for ``models/mvs/mvs_utils.py`` ``construct_vox_points_closest`` (:537-561; called at
run/train_ft_nonstop.py:138-139) and ``construct_vox_points_xyz`` (:503-517; data/scannet_ft_dataset.py:443-444) without
``torch_scatter``: same arguments, same returns (centroids, voxel coordinates in ``torch.unique(dim=0)`` order, index of the
member closest to each centroid), computed by libpnerf_hip.so (csrc/pointinit.hip) deterministically -- the reference's
scatter_mean adds in atomic order, ties of scatter_min are implementation-defined; here sums run in point order and ties go to
the lowest index.  ``partition_xyz`` (voxelise by one cloud, average another) is not used by any reference script and raises."""
import ctypes

import torch

from . import _lib as L


def _space(xyz, vox_res, space_min, space_max):
    """The reference's fp32 arithmetic for the cube that is voxelised (mvs_utils.py:541-551)."""
    if space_min is None:
        xyz_min, xyz_max = torch.min(xyz, dim=-2)[0], torch.max(xyz, dim=-2)[0]
        space_edge = torch.max(xyz_max - xyz_min) * 1.05
        xyz_mid = (xyz_max + xyz_min) / 2
        space_min = xyz_mid - space_edge / 2
        vox = (space_edge / vox_res).expand(3)
    else:
        space_min = torch.as_tensor(space_min, dtype=torch.float32, device=xyz.device)
        space_max = torch.as_tensor(space_max, dtype=torch.float32, device=xyz.device)
        vox = (space_max - space_min) / vox_res
    return space_min.float().cpu(), vox.float().cpu()


def _downsample(xyz_val, vox_res, partition_xyz, space_min, space_max):
    if partition_xyz is not None:
        raise NotImplementedError("partition_xyz is outside the hot-path scope (no reference script passes it)")
    if not xyz_val.is_cuda:
        raise RuntimeError("pointnerf_amd: point initialisation runs on the device only (no CPU implementation)")
    xyz = xyz_val.detach().reshape(-1, 3).contiguous().float()
    n, res = xyz.shape[0], int(vox_res)
    smin, vox = _space(xyz, res, space_min, space_max)
    lib = L.lib()
    nws = lib.pnerf_voxel_downsample_workspace_bytes(n, res, res, res)
    if nws == 0:
        raise RuntimeError("pointnerf_amd: vox_res=%d is too fine for 32-bit voxel keys" % res)
    dev = xyz.device
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    cen = torch.empty(n, 3, dtype=torch.float32, device=dev)
    gidx = torch.empty(n, 3, dtype=torch.int32, device=dev)
    midx = torch.empty(n, dtype=torch.int64, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    fa = (ctypes.c_float * 3)
    L.check(lib.pnerf_voxel_downsample(ctypes.c_void_p(xyz.data_ptr()), n, fa(*smin.tolist()), fa(*vox.tolist()), res, res, res,
                                       ctypes.c_void_p(cen.data_ptr()), ctypes.c_void_p(gidx.data_ptr()), ctypes.c_void_p(midx.data_ptr()),
                                       ctypes.c_void_p(counts.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nws,
                                       ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "pnerf_voxel_downsample")
    m, outside = (int(v) for v in counts.cpu())
    return cen[:m], gidx[:m], midx[:m], outside


def construct_vox_points_closest(xyz_val, vox_res, partition_xyz=None, space_min=None, space_max=None):
    """-> (xyz_centroid [M,3] f32, sparse_grid_idx [M,3] i32, min_idx [M] i64); mvs_utils.py:537-561."""
    cen, gidx, midx, _ = _downsample(xyz_val, vox_res, partition_xyz, space_min, space_max)
    return cen, gidx, midx


def construct_vox_points_xyz(xyz_val, vox_res, partition_xyz=None, space_min=None, space_max=None):
    """-> xyz_centroid [M,3]; mvs_utils.py:503-517."""
    return _downsample(xyz_val, vox_res, partition_xyz, space_min, space_max)[0]
