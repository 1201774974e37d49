"""Checkpoint and visualisation wire formats either side of the path -- host-side only, no kernels.

* `load_seer_checkpoint`: the directory layout `accelerator.save_state` writes and `inference_img.py:98-104` reads
  (`pytorch_model.bin` = SeerUNet, `pytorch_model_1.bin` = FSTextTransformer, both `load_state_dict(strict=True)`).
* `save_visualization_onegif`: the GIF + PNG grid of `utils/ddim_sampling_utils.py:95-123` (conditioning frames followed by
  the sampled frames, 2-pixel black frame around every clip, fps 4; grid = [cond | green bar | prediction | red bar], padded
  by 6 pixels of 0.5 grey like torchvision's make_grid(nrow=1, padding=6, pad_value=0.5)).  The reference encodes with
  imageio / torchvision (neither is a dependency here): frames and grid are built with the same arithmetic and written with
  PIL, so the PIXELS are the reference's; the container bytes of the GIF / PNG encoders differ.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F


def load_seer_checkpoint(load_path: str, sunet=None, fstext_model=None) -> Tuple[object, object]:
    """inference_img.py:98-104: strict load of both state dicts from an `accelerator.save_state` directory."""
    if sunet is not None:
        sunet.load_state_dict(torch.load(os.path.join(load_path, "pytorch_model.bin"), map_location="cpu"), strict=True)
    if fstext_model is not None:
        fstext_model.load_state_dict(torch.load(os.path.join(load_path, "pytorch_model_1.bin"), map_location="cpu"), strict=True)
    return sunet, fstext_model


def gif_frames(x_samples_ddim: torch.Tensor, x0_image: torch.Tensor, num_sample_rows: int = 1) -> np.ndarray:
    """[b,3,f,H,W] samples and [b,3,f0,H,W] conditioning frames, both in [0,1] -> uint8 [f0+f, rows*(H+4), cols*(W+4), 3]
    (ddim_sampling_utils.py:100-104: F.pad 2, cat over frames, '(i j) c f h w -> c f (i h) (j w)', *255 truncated)."""
    vids = F.pad(x_samples_ddim.detach().float().cpu().contiguous(), (2, 2, 2, 2))
    cond = F.pad(x0_image.detach().float().cpu().contiguous(), (2, 2, 2, 2))
    allv = torch.cat([cond, vids], dim=2)                                  # b c f h w
    b, c, f, h, w = allv.shape
    i = num_sample_rows
    assert b % i == 0, "batch must split into num_sample_rows rows"
    j = b // i
    one = allv.reshape(i, j, c, f, h, w).permute(2, 3, 0, 4, 1, 5).reshape(c, f, i * h, j * w)
    return (one.permute(1, 2, 3, 0).numpy() * 255).astype("uint8")


def image_grid(x_samples_ddim: torch.Tensor, x0_image: torch.Tensor) -> torch.Tensor:
    """the PNG grid of ddim_sampling_utils.py:109-121 as a float [3, Hg, Wg] tensor in [0,1]"""
    vids = F.pad(x_samples_ddim.detach().float().cpu().contiguous(), (2, 2, 2, 2))
    cond = F.pad(x0_image.detach().float().cpu().contiguous(), (2, 2, 2, 2))
    flat = lambda t: t.permute(0, 1, 3, 2, 4).reshape(t.shape[0], t.shape[1], t.shape[3], t.shape[2] * t.shape[4])  # b c h (f w)
    pred, cnd = flat(vids), flat(cond)
    n, c, h = pred.shape[0], pred.shape[1], pred.shape[2]
    red, green = torch.ones(n, c, h, 4), torch.ones(n, c, h, 4)
    red[:, [1, 2]] = 0
    green[:, [0, 2]] = 0
    data = torch.cat([cnd, green, pred, red], dim=-1)
    return _make_grid_rows(data)      # torchvision.utils.make_grid(data, nrow=1, padding=6, pad_value=0.5)


def save_visualization_onegif(x_samples_ddim: torch.Tensor, x0_image: torch.Tensor, sample_id: int, image_path: str,
                              num_sample_rows: int = 1) -> Tuple[str, str]:
    """writes `<image>_<id>.gif` (fps 4) and `<image>_grid_<id>.png`; returns both paths"""
    from PIL import Image
    base = image_path.rsplit(".", 1)[0]
    frames = gif_frames(x_samples_ddim, x0_image, num_sample_rows)
    gif_path = f"{base}_{int(sample_id)}.gif"
    imgs = [Image.fromarray(fr) for fr in frames]
    imgs[0].save(gif_path, save_all=True, append_images=imgs[1:], duration=250, loop=0)
    grid = image_grid(x_samples_ddim, x0_image)
    png = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()     # torchvision.utils.save_image
    png_path = f"{base}_grid_{int(sample_id)}.png"
    Image.fromarray(png).save(png_path)
    return gif_path, png_path


def _make_grid_rows(data: torch.Tensor, pad: int = 6, pad_value: float = 0.5) -> torch.Tensor:
    """torchvision.utils.make_grid(data, nrow=1, padding=pad, pad_value=pad_value): one image per row, a border of `pad_value`
    around the grid and between the rows -- except for n == 1, which make_grid returns unpadded"""
    n, c, h, W = data.shape
    if n == 1:              # torchvision: a single image comes back as it is (tensor.squeeze(0)), no border
        return data[0].clone()
    grid = torch.full((c, n * (h + pad) + pad, W + 2 * pad), pad_value)
    for k in range(n):
        y0 = pad + k * (h + pad)
        grid[:, y0:y0 + h, pad:pad + W] = data[k]
    return grid


def _write_gif(frames: np.ndarray, path: str) -> None:
    from PIL import Image
    imgs = [Image.fromarray(fr) for fr in frames]
    imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=250, loop=0)       # imageio.mimwrite(..., fps=4)


def _write_png(grid: torch.Tensor, path: str) -> None:
    from PIL import Image
    png = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()     # torchvision.utils.save_image
    Image.fromarray(png).save(path)


def validation_grid(x_samples_ddim: torch.Tensor, video_recon: torch.Tensor, video: torch.Tensor) -> torch.Tensor:
    """the PNG grid of ddim_sampling_utils.py:46-92 (train.py / inference.py validation): per sample one row
    [cond x3 stacked | green bar | original over VAE reconstruction over prediction | red bar]; all inputs in [0,1],
    `video` = [b,3,f0+f,H,W] (conditioning frames first), the other two [b,3,f,H,W]"""
    f = x_samples_ddim.shape[2]
    f0 = video.shape[2] - f
    padf = lambda t: F.pad(t.detach().float().cpu().contiguous(), (2, 2, 2, 2))
    flat = lambda t: t.permute(0, 1, 3, 2, 4).reshape(t.shape[0], t.shape[1], t.shape[3], t.shape[2] * t.shape[4])  # b c h (f w)
    pred, recon = flat(padf(x_samples_ddim)), flat(padf(video_recon))
    ori, cond = flat(padf(video[:, :, f0:])), flat(padf(video[:, :, :f0]))
    rows3 = torch.cat([ori, recon, pred], dim=-2)
    cond3 = cond.repeat(1, 1, 3, 1)
    n, c, h = rows3.shape[:3]
    red, green = torch.ones(n, c, h, 4), torch.ones(n, c, h, 4)
    red[:, [1, 2]] = 0
    green[:, [0, 2]] = 0
    return _make_grid_rows(torch.cat([cond3, green, rows3, red], dim=-1))


def save_visualization(vae, x_samples_ddim: torch.Tensor, video_latent: torch.Tensor, video: torch.Tensor,
                       results_folder: str, global_step: int, num_sample_rows: int = 2, gather=None) -> Tuple[str, str, str]:
    """ddim_sampling_utils.py:46-92: `<step>.gif` (samples), `ori_<step>.gif` (ground truth), `image_grid_<step>.png`.
    `video` is in [-1,1] with the conditioning frames first, `video_latent` the scaled latents of its predicted frames
    (decoded here through `vae` for the reconstruction row); `gather` = accelerator.gather or None."""
    g = gather if gather is not None else (lambda t: t)
    f = video_latent.shape[2]
    n = video_latent.shape[0]
    z = video_latent.permute(0, 2, 1, 3, 4).reshape(n * f, *video_latent.shape[1:2], *video_latent.shape[3:]) * (1 / 0.18215)
    rec = vae.decode(z).sample
    rec = (rec.reshape(n, f, *rec.shape[1:]).permute(0, 2, 1, 3, 4) + 1.0) / 2.0
    f0 = video.shape[2] - f
    vid01 = (video + 1.0) / 2.0
    samples, rec, vid01 = g(x_samples_ddim.contiguous()), g(rec.contiguous()), g(vid01.contiguous())
    empty = vid01[:, :, :0]
    os.makedirs(results_folder, exist_ok=True)
    p_gif = os.path.join(results_folder, f"{global_step}.gif")
    p_ori = os.path.join(results_folder, f"ori_{global_step}.gif")
    p_png = os.path.join(results_folder, "image_grid_{}.png".format(int(global_step)))
    _write_gif(gif_frames(samples, empty, num_sample_rows), p_gif)
    _write_gif(gif_frames(vid01[:, :, f0:], empty, num_sample_rows), p_ori)
    _write_png(validation_grid(samples, rec, vid01), p_png)
    return p_gif, p_ori, p_png
