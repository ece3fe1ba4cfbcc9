"""Full-image evaluation on top of the hot path -- what ``test()`` of the reference does around
``model.test()`` (run/train_ft.py:252-414): slice the H x W ray grid into chunks, render each chunk, scatter
``coarse_raycolor`` (after ``fill_invalid``) into a canvas by ``pixel_idx``, score PSNR.

Differences in HOW: the reference renders <= 48^2 = 2304 rays per call (278 calls, 278 voxel-grid rebuilds and a
device->host copy of every chunk for an 800^2 image); here a chunk is as large as the dense query buffers allow
(default 160 000 rays: 4 calls per 800^2 image), the grid is built once, and the canvas and the PSNR stay on the device.
"""
import math

import torch

from .neural_points_volumetric_model import fill_invalid


def pixel_grid(h, w, device):
    """pixel_idx [1, h*w, 2] (px, py), row-major like the datasets' no_crop grid (nerf_synth360_ft_dataset.py:598-613)."""
    py, px = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack([px.reshape(-1), py.reshape(-1)], dim=-1)[None].float()


def rays_from_pixels(pixel_idx, intrinsic, camrotc2w):
    """get_dtu_raydir with dir_norm=False (data/data_utils.py:55-70) on the device: [1,R,2] -> [1,R,3]."""
    x = (pixel_idx[..., 0] + 0.5 - intrinsic[0, 2]) / intrinsic[0, 0]
    y = (pixel_idx[..., 1] + 0.5 - intrinsic[1, 2]) / intrinsic[1, 1]
    dirs = torch.stack([x, y, torch.ones_like(x)], dim=-1)
    return dirs @ camrotc2w[0].T


@torch.no_grad()
def render_image(model, campos, camrotc2w, intrinsic, h, w, near, far, bg_color, chunk=160000, products=None):
    """Returns (image [h, w, 3] on the device, ray_mask [h*w] bool).  ``products`` = 2 renders with two MFMA products per multiply-add
    (ops.set_inference_products: ~1.5x less matrix work, ray colour within ~2e-5 of the three-product render; the previous setting is
    restored on return); None keeps the library's current setting."""
    from . import ops
    if products is not None:
        prev = ops.set_inference_products(products)
        try:
            return render_image(model, campos, camrotc2w, intrinsic, h, w, near, far, bg_color, chunk=chunk)
        finally:
            ops.set_inference_products(prev)
    dev = campos.device
    pix = pixel_grid(h, w, dev)
    canvas = torch.empty(h * w, 3, device=dev)
    hit = torch.empty(h * w, dtype=torch.bool, device=dev)
    for k in range(0, h * w, chunk):
        pi = pix[:, k:k + chunk]
        raydir = rays_from_pixels(pi, intrinsic.to(dev), camrotc2w)
        out = model(campos=campos, raydir=raydir, camrotc2w=camrotc2w, pixel_idx=pi, near=near, far=far, bg_color=bg_color,
                    h=h, w=w, intrinsic=intrinsic)
        full = fill_invalid(out, bg_color)
        canvas[k:k + chunk] = full["coarse_raycolor"][0]
        hit[k:k + chunk] = out["ray_mask"][0] > 0
    return canvas.view(h, w, 3), hit


def psnr(img, gt):
    """mse2psnr of the reference (utils/visualizer.py:140-155): -10 log10(mse)."""
    mse = torch.mean((img.reshape(-1, 3) - gt.reshape(-1, 3).to(img.device)) ** 2)
    return -10.0 * torch.log10(mse)


@torch.no_grad()
def test_views(model, views, opt, height, width, test_num_step=1, chunk=160000, on_view=None):
    """``test()`` of the training / evaluation scripts (run/train_ft.py:252-414, run/test_ft.py:134-274) around the model
    shell: for every ``test_num_step``-th view render all its rays through ``model.set_input / model.test()``, scatter the
    visuals into H x W canvases by ``pixel_idx``, score the items of ``opt.test_color_loss_items``
    (``coarse_raycolor``: MSE of the whole canvas against the ground truth scattered the same way, pixels the view has no
    ray for are 0 in both; ``ray_masked_coarse_raycolor``: MSE over the rays that hit the cloud) and accumulate
    ``mse2psnr`` per view like the Visualizer does (utils/visualizer.py:142-156).

    Returns (psnr of ``opt.test_color_loss_items[0]`` averaged over the views -- the function's return value in the
    reference --, {item: mean loss, item + "_psnr": mean psnr}).  Canvases stay on the device; ``on_view(i, visuals)`` receives
    them (the reference writes PNGs there and afterwards re-reads them for SSIM / LPIPS, which need third-party packages and
    are outside this path).  The chunk is 160 000 rays instead of <= 48^2."""
    model.eval()
    dev = model.device
    items = list(getattr(opt, "test_color_loss_items", ["coarse_raycolor"]))
    acc, count = {}, 0
    for i in range(0, len(views), test_num_step):
        view = views[i]
        raydir = view["raydir"].to(dev)
        pixel_idx = view["pixel_idx"].to(dev)
        pixel_idx = pixel_idx.reshape(pixel_idx.shape[0], -1, pixel_idx.shape[-1])
        total = pixel_idx.shape[1]
        pl = pixel_idx[0].to(torch.long)
        edge = torch.zeros([height, width], dtype=torch.bool, device=dev)
        edge[pl[:, 1], pl[:, 0]] = True
        visuals, ray_masks = {}, []
        for k in range(0, total, chunk):
            data = {kk: vv for kk, vv in view.items() if kk != "gt_mask"}
            data["raydir"] = raydir[:, k:k + chunk, :]
            data["pixel_idx"] = pixel_idx[:, k:k + chunk, :]
            data["gt_image"] = view["gt_image"][:, k:k + chunk, :]
            model.set_input(data)
            model.test()
            pid = pl[k:k + chunk]
            for key, value in model.get_current_visuals(data=data).items():
                if value is None or key == "gt_image":
                    continue
                if key not in visuals:
                    visuals[key] = torch.zeros((height, width, 3), dtype=value.dtype, device=dev)
                visuals[key][pid[:, 1], pid[:, 0], :] = value[0]
            ray_masks.append(model.output["ray_mask"] > 0)
        ray_masks = torch.cat(ray_masks, dim=1)                                   # [1, P]
        gt_rays = view["gt_image"].to(dev).reshape(-1, 3)
        gt_canvas = torch.zeros((height * width, 3), dtype=torch.float32, device=dev)
        gt_canvas[edge.reshape(-1)] = gt_rays            # rays must come in row-major pixel order (the datasets' no_crop grid), :332-333
        visuals["gt_image"] = gt_canvas.reshape(height, width, 3)
        losses = {}
        if "coarse_raycolor" in items:
            losses["coarse_raycolor"] = torch.mean((visuals["coarse_raycolor"].reshape(-1, 3) - gt_canvas) ** 2)
        if "ray_masked_coarse_raycolor" in items:
            pred = visuals["coarse_raycolor"][pl[:, 1], pl[:, 0]][ray_masks[0]]
            losses["ray_masked_coarse_raycolor"] = torch.mean((pred - gt_rays[ray_masks[0]]) ** 2)
        for kk, vv in losses.items():
            acc[kk] = acc.get(kk, 0) + vv
            acc[kk + "_psnr"] = acc.get(kk + "_psnr", 0) + (-10.0 * torch.log(vv) / math.log(10.0))
        count += 1
        if on_view is not None:
            on_view(i, visuals)
    avg = {k: float(v) / max(count, 1) for k, v in acc.items()}
    return avg.get(items[0] + "_psnr"), avg

