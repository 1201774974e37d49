"""The caller harness of the path -- our counterpart of `inference_img.py:164-187` over tensors.

The reference script loads an image and a prompt, runs CLIP, and then does exactly this with the results; the HF
downloads, PIL / GIF I/O and the CLIP tokenizer / text encoder are not reproduced (their outputs are the inputs here):

    x0_image  [b, 3, f1, H, W] in [-1, 1]  --vae.encode(.).latent_dist.sample() * 0.18215-->  x0_emb [b, 4, f1, H/8, W/8]
    text_emb  [b, 77, 768] (CLIP of the prompt)  --FSTextTransformer-->  c  [b, F, 77, 768]
    empty_emb [b, 77, 768] (CLIP of '')          --unsqueeze(1).expand-->  uc [b, F, 77, 768]   (same frame dim as c: the
                                                                           batched-CFG branch of ddim_video.py:200-204)
    noise = torch.randn(b, 4, F - f1, h, w) drawn on the CPU generator, then moved (inference_img.py:179), redrawn after
    every sample (:187);  `num_samples` sequential ddim_sample calls  ->  clips [b, 3, F - f1, H, W] in [0, 1]

Everything numeric runs in libseer_hip.so through SeerUNet / FSTextTransformer / AutoencoderKL / DDIMSampler.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .ddim import ddim_sample


@torch.no_grad()
def generate_clips(sunet, fstext_model, vae, sampler, x0_image: torch.Tensor, text_emb: torch.Tensor,
                   empty_emb: torch.Tensor, *, num_frames: int, cond_frames: int, ddim_steps: int = 30,
                   scale: float = 7.5, num_samples: int = 1, noise_generator: Optional[torch.Generator] = None,
                   latent_generator: Optional[torch.Generator] = None) -> List[torch.Tensor]:
    """x0_image: [b, 3, 1, H, W] (one image, repeated over the conditioning frames like inference_img.py:166) or
    [b, 3, cond_frames, H, W].  Returns `num_samples` clips [b, 3, num_frames - cond_frames, H, W] in [0, 1]."""
    dev = x0_image.device
    f1, f2 = cond_frames, num_frames - cond_frames
    if x0_image.shape[2] == 1:
        x0_image = x0_image.expand(-1, -1, f1, -1, -1)
    assert x0_image.shape[2] == f1, "x0_image must hold one frame or cond_frames frames"
    b = x0_image.shape[0]
    frames = x0_image.permute(0, 2, 1, 3, 4).reshape(b * f1, *x0_image.shape[1:2], *x0_image.shape[3:])    # (b f) c h w
    lat = vae.encode(frames).latent_dist.sample(generator=latent_generator) * 0.18215
    x0_emb = lat.reshape(b, f1, *lat.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()                       # b c f h w
    _, c_l, _, h_l, w_l = x0_emb.shape

    fstext_model.set_numframe(num_frames)
    c = fstext_model(context=text_emb)
    uc = empty_emb.unsqueeze(1).expand(-1, c.shape[1], -1, -1).contiguous()

    clips = []
    for _ in range(num_samples):
        noise = torch.randn((b, c_l, f2, h_l, w_l), generator=noise_generator).to(dev)     # CPU draw, then moved (:179,187)
        clips.append(ddim_sample(sampler, sunet, vae, shape=(b, c_l, f2, h_l, w_l), c=c, start_code=noise, x0_emb=x0_emb,
                                 ddim_steps=ddim_steps, scale=scale, uc=uc))
    return clips


def concat_all_gather(t: torch.Tensor, process_group=None) -> torch.Tensor:
    """eval.py's `concat_all_gather(accelerator, t)` (= accelerator.gather): the per-rank batches stacked in rank order"""
    import torch.distributed as dist
    # None = the default (WORLD) group, as accelerator.gather always gathers over WORLD (eval.py:226-231)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return t
    t = t.contiguous()
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(process_group))]
    dist.all_gather(parts, t, group=process_group)
    return torch.cat(parts, 0)


@torch.no_grad()
def evaluate_batch(sunet, fstext_model, vae, sampler, video: torch.Tensor, text_emb: torch.Tensor, empty_emb: torch.Tensor, *,
                   cond_frames: int, ddim_steps: int = 30, scale: float = 7.5, process_group=None, gather: bool = True,
                   noise_generator: Optional[torch.Generator] = None, latent_generator: Optional[torch.Generator] = None):
    """The body of eval.py's validation loop (eval.py:186-231) for THIS rank's batch: the first `cond_frames` frames of
    `video` [b, 3, F, H, W] in [-1, 1] condition the rest; returns (pred, gt) = ([conditioning frames | sampled frames],
    ground truth), both [N*b, 3, F, H, W] in [0, 1] gathered over the ranks of `process_group` (None = every rank, as
    accelerator.gather does) in rank order.  N GPUs evaluate N batches with no communication until this final gather (the
    samples are the independent units of the path); gather=False returns this rank's batch only."""
    x0 = video[:, :, :cond_frames]
    clip = generate_clips(sunet, fstext_model, vae, sampler, x0, text_emb, empty_emb, num_frames=video.shape[2],
                          cond_frames=cond_frames, ddim_steps=ddim_steps, scale=scale, num_samples=1,
                          noise_generator=noise_generator, latent_generator=latent_generator)[0]
    pred = torch.cat([(x0.float() + 1.0) / 2.0, clip], dim=2)
    gt = (video.float() + 1.0) / 2.0
    if not gather:
        return pred, gt
    return concat_all_gather(pred, process_group), concat_all_gather(gt, process_group)
