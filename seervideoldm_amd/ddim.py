"""DDIMSampler / ddim_sample -- host-side mirror of the reference's sampler surface
(ldm/models/diffusion/ddim_video.py:14-238, utils/ddim_sampling_utils.py:21-42).

Same names, keyword arguments and return values.  Differences that are not semantic:
  * the schedule tables are computed on the host exactly like the reference (float64 -> float32 values) and uploaded
    ONCE as a [S', 4] fp32 table; a step reads row `index` on the device, so there are no per-step host->device scalar
    copies (the reference does four `torch.full(numpy_scalar)` per step, ddim_video.py:219-222);
  * CFG combine + DDIM update are one fused HIP kernel (seer_cfg_ddim_step);
  * `register_buffer` does not hard-code "cuda" (SURVEY finding 5);
  * the batched-CFG inputs [uc, c] are concatenated once per `sample()` call, not once per step.
The reference draws `torch.randn(x.shape)` every step even when sigma == 0 (ddim_video.py:234); we keep the draw so a seeded
multi-sample run consumes the device RNG stream identically.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import ops


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """ldm/modules/diffusionmodules/util.py:21-44 ('linear' is sqrt-space linspace; the only one the path uses)."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """util.py:46-60 -- note c = T // S, so S=30 yields 31 timesteps."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


class DDIMSampler(object):
    def __init__(self, device, timesteps=1000, schedule="linear", **kwargs):
        self.ddpm_num_timesteps = timesteps
        self.schedule = schedule
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.consume_rng_when_deterministic = True

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, given_betas=None, beta_schedule="linear", timesteps=1000,
                      linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, ddim_discretize="uniform", ddim_eta=0.,
                      verbose=True):
        """ddim_video.py:27-68.  Host arithmetic mirrors the reference's dtypes: cumprod in float64, stored float32."""
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas_cumprod = np.cumprod(1. - betas, axis=0)
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        ac32 = torch.tensor(alphas_cumprod, dtype=torch.float32)
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = ac32
        self.alphas_cumprod_prev = torch.tensor(np.append(1., alphas_cumprod[:-1]), dtype=torch.float32)
        ts = self.ddim_timesteps
        a32 = ac32[ts].numpy()
        self.ddim_alphas = a32.astype(np.float64)
        self.ddim_alphas_prev = np.asarray([float(ac32[0])] + ac32[ts[:-1]].tolist())
        self.ddim_sigmas = ddim_eta * np.sqrt((1 - self.ddim_alphas_prev) / (1 - self.ddim_alphas) *
                                              (1 - self.ddim_alphas / self.ddim_alphas_prev))
        self.ddim_sqrt_one_minus_alphas = np.sqrt(np.float32(1.0) - a32).astype(np.float64)
        coef = np.stack([self.ddim_alphas, self.ddim_alphas_prev, self.ddim_sigmas, self.ddim_sqrt_one_minus_alphas], 1)
        self.ddim_coef = torch.tensor(coef, dtype=torch.float32, device=self.device)     # ONE upload per schedule
        self._t_table = torch.tensor(ts, dtype=torch.long, device=self.device)

    @torch.no_grad()
    def sample(self, unet, S, batch_size, shape, x0_emb=None, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, eta=0., mask=None, x0=None, cond_frames=0, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, null_cond_prob=None, is_3d=False,
               **kwargs):
        if conditioning is not None and not isinstance(conditioning, dict):
            if conditioning.shape[0] != batch_size:
                print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        if is_3d:
            C, Fr, H, W = shape
            size = (batch_size, C, Fr, H, W)
        else:
            raise NotImplementedError("the Seer hot path is 5-D (is_3d=True, ddim_sampling_utils.py:36)")
        return self.ddim_sampling(unet, conditioning, size, x0_emb=x0_emb, is_3d=is_3d, callback=callback,
                                  img_callback=img_callback, cond_frames=cond_frames, temperature=temperature,
                                  noise_dropout=noise_dropout, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @torch.no_grad()
    def ddim_sampling(self, unet, cond, shape, is_3d, x0_emb=None, cond_frames=0, x_T=None, callback=None,
                      img_callback=None, log_every_t=100, temperature=1., noise_dropout=0.,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        device = self.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        # one clip sharded over several ranks (parallel.attach): every rank must denoise the SAME latent.  The start code,
        # the conditioning latents (a per-rank posterior sample) and any per-step noise come from rank 0.
        img = _from_rank0(unet, img)
        if x0_emb is not None:
            x0_emb = _from_rank0(unet, x0_emb.to(device))
        timesteps = self.ddim_timesteps
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        total_steps = timesteps.shape[0]
        for i, step in enumerate(np.flip(timesteps)):
            index = total_steps - i - 1
            ts = self._t_table[index].expand(b)
            img, pred_x0 = self.p_sample_ddim(unet, img, cond, ts, index=index, is_3d=is_3d, x0_emb=x0_emb,
                                              cond_frames=cond_frames, temperature=temperature,
                                              noise_dropout=noise_dropout,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, unet, x, c, t, index, is_3d=True, x0_emb=None, cond_frames=0, repeat_noise=False,
                      use_original_steps=False, temperature=1., noise_dropout=0., score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1., unconditional_conditioning=None,
                      null_cond_prob=None):
        """ddim_video.py:183-238."""
        if use_original_steps or noise_dropout > 0. or repeat_noise:
            raise NotImplementedError("only the DDIM-subsequence path that ddim_sample drives is built")
        b = x.shape[0]
        cond_f = 0
        x = x.to(torch.float32).contiguous()
        x_cat = x
        if x0_emb is not None:
            cond_f = x0_emb.shape[2]
            x_cat = torch.cat([x0_emb.to(x.dtype), x], dim=2)
        uc, scale = unconditional_conditioning, unconditional_guidance_scale
        t = t.to(torch.long)
        if uc is None or scale == 1.:
            eps = unet(x_cat, t, c)
            cfg = False
        elif uc.shape[2] == c.shape[2]:
            cached = getattr(self, "_cfg_inputs", None)
            if (cached is None or cached[0] is not c or cached[1] is not uc or cached[3] != (c._version, uc._version)):
                # one [uc, c] tensor per (c, uc) pair: its identity keys the UNet's cross-attention K/V cache and graph
                cached = (c, uc, torch.cat([uc, c]).contiguous(), (c._version, uc._version))
                self._cfg_inputs = cached
            eps = unet(torch.cat([x_cat] * 2), torch.cat([t] * 2), cached[2], cond_frame=cond_frames)
            cfg = True
        else:
            e_uc = unet(x_cat, t, uc, cond_frame=cond_frames)
            e_c = unet(x_cat, t, c, cond_frame=cond_frames)
            eps = torch.cat([e_uc, e_c])
            cfg = True
        sigma = float(self.ddim_sigmas[index])
        noise = None
        if sigma != 0. or self.consume_rng_when_deterministic:
            noise = torch.randn(x.shape, device=x.device) * temperature
            if sigma != 0.:
                noise = _from_rank0(unet, noise)
        x_prev, pred_x0 = ops.cfg_ddim_step(eps.float().contiguous(), x, self.ddim_coef, index, cfg=cfg, scale=scale,
                                            cond_f=cond_f, noise=noise if sigma != 0. else None)
        return x_prev, pred_x0


def _from_rank0(unet, t: torch.Tensor) -> torch.Tensor:
    """broadcast `t` from rank 0 when `unet` runs one clip sharded over several ranks; identity otherwise"""
    shard = getattr(unet, "_shard", None)
    if shard is None or shard.world <= 1:
        return t
    import torch.distributed as dist
    t = t.contiguous()
    dist.broadcast(t, src=0)
    return t


@torch.no_grad()
def ddim_sample(sampler, unet, vae, shape, c, start_code, x0_emb, ddim_steps=10, scale=1.0, uc=None):
    """utils/ddim_sampling_utils.py:21-42: sampler -> 1/0.18215 -> vae.decode -> clamp((x+1)/2, 0, 1)."""
    if scale == 1.0:
        uc = None
    samples_ddim, _ = sampler.sample(unet=unet, S=ddim_steps, conditioning=c, batch_size=shape[0], shape=shape[1:],
                                     x0_emb=x0_emb, verbose=False, unconditional_guidance_scale=scale,
                                     unconditional_conditioning=uc, eta=0.0, x_T=start_code, is_3d=True)
    n, ch, f, h, w = samples_ddim.shape
    z = (samples_ddim.permute(0, 2, 1, 3, 4).reshape(n * f, ch, h, w) * (1 / 0.18215)).contiguous()
    x = vae.decode(z).sample
    x = x.reshape(n, f, *x.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()
    return ops.clamp01_(x.float())
