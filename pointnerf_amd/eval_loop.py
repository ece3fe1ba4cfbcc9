"""Full-image evaluation on top of the hot path -- what ``test()`` of the reference does around
``model.test()`` (run/train_ft.py:252-414): slice the H x W ray grid into chunks, render each chunk, scatter
``coarse_raycolor`` (after ``fill_invalid``) into a canvas by ``pixel_idx``, score PSNR.

Differences in HOW: the reference renders <= 48^2 = 2304 rays per call (278 calls, 278 voxel-grid rebuilds and a
device->host copy of every chunk for an 800^2 image); here a chunk is as large as the dense query buffers allow
(default 160 000 rays: 4 calls per 800^2 image), the grid is built once, and the canvas and the PSNR stay on the device.
"""
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
def render_image(model, campos, camrotc2w, intrinsic, h, w, near, far, bg_color, chunk=160000):
    """Returns (image [h, w, 3] on the device, ray_mask [h*w] bool)."""
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
