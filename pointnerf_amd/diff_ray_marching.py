"""Drop-in for the two functions of ``models/rendering/diff_ray_marching.py`` that are on the hot path:

  ray_march(ray_dist, ray_valid, ray_features, render_func, blend_func, bg_color=None) -> 7-tuple   (:508-554)
  near_far_linear_ray_generation(campos, raydir, point_count, near, far, jitter)                   (:349-392)

``ray_march`` runs in libpnerf_hip.so (one wavefront per ray, ``pnerf_raymarch_forward/backward``) for the
radiance / alpha configuration every script uses; gradient flows from ``ray_color`` to ``ray_features`` (the other
outputs are returned for inspection, non-differentiable).  The ray generator materialises [N,R,D,3] like the
reference does; the fused querier never calls it (it generates samples in-kernel), it exists for callers of the
reference API such as the native-op tests.
"""
import ctypes

import torch

from . import _lib as L
from . import ops
from .diff_render_func import radiance_render, alpha_blend


class _RayMarch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray_dist, ray_valid, feats, bg):
        R, SR = ray_dist.shape[-2], ray_dist.shape[-1]
        dev = feats.device
        rd = ray_dist.detach().reshape(R, SR).contiguous().float()
        rv = ray_valid.detach().reshape(R, SR).to(torch.uint8).contiguous()
        ft = feats.detach().reshape(R, SR, 4).contiguous().float()
        f32 = dict(dtype=torch.float32, device=dev)
        color, op, acc = torch.empty(R, 3, **f32), torch.empty(R, SR, **f32), torch.empty(R, SR, **f32)
        bw, bgt = torch.empty(R, SR, **f32), torch.empty(R, **f32)
        bg3 = None if bg is None else (ctypes.c_float * 3)(*[float(x) for x in bg.detach().reshape(-1)[:3].cpu().tolist()])
        L.check(L.lib().pnerf_raymarch_forward(ops._ptr(rd), ops._ptr(rv), ops._ptr(ft), bg3, R, SR, ops._ptr(color), ops._ptr(op),
                                               ops._ptr(acc), ops._ptr(bw), ops._ptr(bgt), ops._stream()), "pnerf_raymarch_forward")
        ctx.save_for_backward(rd, rv, ft)
        ctx.bg3, ctx.shape = bg3, tuple(feats.shape)
        ctx.mark_non_differentiable(op, acc, bw, bgt)
        return color, op, acc, bw, bgt

    @staticmethod
    def backward(ctx, g_color, *unused):
        rd, rv, ft = ctx.saved_tensors
        R, SR = rd.shape
        g = g_color.reshape(R, 3).contiguous().float()
        gf = torch.empty(R, SR, 4, dtype=torch.float32, device=g.device)
        L.check(L.lib().pnerf_raymarch_backward(ops._ptr(rd), ops._ptr(rv), ops._ptr(ft), ctx.bg3, R, SR, ops._ptr(g), ops._ptr(gf),
                                                ops._stream()), "pnerf_raymarch_backward")
        return None, None, gf.view(ctx.shape), None


def ray_march(ray_dist, ray_valid, ray_features, render_func, blend_func, bg_color=None):
    """Reference signature and return order: (ray_color [N,R,3], point_color [N,R,S,3], opacity [N,R,S],
    acc_transmission [N,R,S], blend_weight [N,R,S,1], background_transmission [N,R,1], background_blend_weight)."""
    if render_func is not radiance_render or blend_func is not alpha_blend:
        raise NotImplementedError("ray_march: only radiance_render / alpha_blend (every script's setting) run on the HIP path")
    ops._need_cuda(ray_features, "ray_features")
    N, R, S = ray_dist.shape
    assert N == 1, "batch size 1 (every reference script)"
    color, op, acc, bw, bgt = _RayMarch.apply(ray_dist, ray_valid, ray_features, bg_color)
    bgt = bgt.view(N, R, 1)
    return color.view(N, R, 3), radiance_render(ray_features), op.view(N, R, S), acc.view(N, R, S), bw.view(N, R, S, 1), bgt, \
        alpha_blend(1, bgt)


def near_far_linear_ray_generation(campos, raydir, point_count, near=0.1, far=10, jitter=0., **kargs):
    """Reference signature (diff_ray_marching.py:349-392): returns (raypos [N,R,S,3], segment_length [N,R,S], valid,
    middle_point_ts).  Depth range split into S equal segments, optionally jittered by +-jitter/2 of their length,
    samples at the segment mid-points; ``raydir`` is used un-normalised, so depths are camera-z depths."""
    dev = campos.device
    N, R = raydir.shape[0], raydir.shape[1]
    u = torch.linspace(0, 1, point_count + 1, device=dev).view(1, -1)
    edges = near * (1 - u) + far * u
    seg = edges[..., 1:] - edges[..., :-1]
    noise = torch.rand((N, R, point_count), device=dev) - 0.5
    seg = seg * (1 + jitter * noise)                                   # [N,R,S]
    ends = near + torch.cat([seg.new_zeros(N, R, 1), torch.cumsum(seg, dim=2)], dim=2)
    mid = 0.5 * (ends[..., :-1] + ends[..., 1:])
    raypos = campos[:, None, None, :] + raydir[:, :, None, :] * mid[..., None]
    seg_len = seg * torch.linalg.norm(raydir, dim=-1, keepdim=True)
    return raypos, seg_len, torch.ones_like(mid), mid
