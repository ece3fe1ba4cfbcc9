"""Helpers for the -m gpu tests: run the HIP path (through the C ABI) on a seeded case."""
import torch

from pointnerf_amd import ops
from pointnerf_amd.point_query import lighting_fast_querier

DEV = "cuda:0"


def hip_render(opt, xyz, attrs, inp, mlp, train=False):
    """query_dense + render_forward on the device.  Returns (dense, fwd, ctx) with ctx holding what backward needs."""
    dev = torch.device(DEV)
    xyz_d = xyz.to(dev).contiguous()
    pts_t = {k: v.detach().to(dev).reshape(v.shape[1], v.shape[2]).contiguous() for k, v in attrs.items()}
    qr = lighting_fast_querier(dev, opt)
    raydir = inp["raydir"][0].to(dev).contiguous()
    dense = qr.query_dense(xyz_d[None], xyz.shape[0], float(inp["near"].min()), float(inp["far"].max()), raydir[None], inp["campos"].to(dev))
    n_valid = int(dense["counters"][0].item())
    flat = ops.flatten_mlp(mlp, dev)
    packed = ops.pack_mlp(flat)
    cam = ops.make_camera(inp["campos"][0].numpy(), inp["camrotc2w"][0].numpy(), opt.vsize[2], opt.raydist_mode_unit,
                          bg=inp["bg_color"][0].numpy())
    pts = ops.make_points(xyz_d, pts_t["points_embeding"], pts_t["points_conf"], pts_t["points_dir"], pts_t["points_color"])
    R = raydir.shape[0]
    fwd = ops.render_forward(cam, pts, packed, flat, raydir, dense, R, opt.SR, opt.K, n_valid, train)
    ctx = dict(cam=cam, pts=pts, pts_t=pts_t, xyz=xyz_d, packed=packed, flat=flat, raydir=raydir, R=R, n_valid=n_valid)
    return dense, fwd, ctx
