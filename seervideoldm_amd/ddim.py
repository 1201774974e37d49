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


def _betas(n_train: int, beta_first: float, beta_last: float) -> np.ndarray:
    """The one noise schedule of this path: sqrt(beta) rises in n_train equal steps from sqrt(beta_first) to
    sqrt(beta_last); float64, like the reference's table (ldm/modules/diffusionmodules/util.py:23-26 under the name "linear")."""
    root = torch.linspace(float(beta_first) ** 0.5, float(beta_last) ** 0.5, int(n_train), dtype=torch.float64)
    return (root * root).numpy()


def _strided_timesteps(n_sample: int, n_train: int) -> np.ndarray:
    """Every (n_train // n_sample)-th training step, shifted by one: S = 50 -> 1, 21, ..., 981; S = 30 has stride 33 and
    therefore 31 entries (util.py:46-60, the "uniform" rule)."""
    stride = int(n_train) // int(n_sample)
    if stride < 1:
        raise ValueError(f"{n_sample} sampling steps do not fit into {n_train} training steps")
    return np.arange(0, int(n_train), stride, dtype=np.int64) + 1


class DDIMSampler(object):
    def __init__(self, device, timesteps=1000, schedule="linear", **kwargs):
        self.ddpm_num_timesteps = timesteps
        self.schedule = schedule
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.consume_rng_when_deterministic = True
        # p_sample_ddim under graph replay hands out its static latent buffers instead of copies of them: the next step
        # overwrites what the previous one returned.  ddim_sampling sets it for its own loop (and copies what it keeps).
        self.static_step_outputs = False

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, given_betas=None, beta_schedule="linear", timesteps=1000,
                      linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, ddim_discretize="uniform", ddim_eta=0.,
                      verbose=True):
        """ddim_video.py:27-68.  Host arithmetic mirrors the reference's dtypes: cumprod in float64, stored float32."""
        if ddim_discretize != "uniform" or (given_betas is None and beta_schedule != "linear"):
            # the sampler scripts never ask for anything else (ddim_sampling_utils.py:29-36); the reference's other branches
            # are not on this path
            raise NotImplementedError(f"schedule {beta_schedule!r} / discretisation {ddim_discretize!r}: this path has the "
                                      "'linear' (sqrt-space) betas and the 'uniform' stride only")
        self.ddim_timesteps = _strided_timesteps(ddim_num_steps, self.ddpm_num_timesteps)
        if verbose:
            print(f"DDIM timesteps ({len(self.ddim_timesteps)}): {self.ddim_timesteps}")
        betas = np.asarray(given_betas, dtype=np.float64) if given_betas is not None else _betas(timesteps, linear_start, linear_end)
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
        # inside this loop a captured step may hand out its static buffers (nothing here keeps a step's output past the next
        # step without copying it)
        static_before, self.static_step_outputs = self.static_step_outputs, True
        try:
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
                    img_callback(pred_x0.clone(), i)
                if index % log_every_t == 0 or index == total_steps - 1:
                    intermediates["x_inter"].append(img.clone())
                    intermediates["pred_x0"].append(pred_x0.clone())
        finally:
            self.static_step_outputs = static_before
        return img.clone(), intermediates

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
        uc, scale = unconditional_conditioning, unconditional_guidance_scale
        sigma = float(self.ddim_sigmas[index])
        # the whole step -- input assembly, CFG-batched UNet, CFG combine, DDIM update -- as ONE replayed hipGraph when the model
        # replays graphs anyway (unet.use_graph) and the step is the one ddim_sample drives: batched CFG or none, eta = 0, and
        # t = the schedule's own timestep of `index` (recognised by its address: a view of the device timestep table)
        if (sigma == 0. and getattr(unet, "use_graph", False) and x.is_cuda and (uc is None or scale == 1. or uc.shape[2] == c.shape[2])
                and torch.is_tensor(t) and t.dtype == torch.long and t.data_ptr() == self._t_table.data_ptr() + 8 * int(index)):
            plain = uc is None or scale == 1.
            done = self._graph_step(unet, x, c, None if plain else uc, index, x0_emb, 0 if plain else cond_frames, scale)
            if done is not None:
                return done
        x_cat = x
        if x0_emb is not None:
            cond_f = x0_emb.shape[2]
            x_cat = torch.cat([x0_emb.to(x.dtype), x], dim=2)
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
        noise = None
        if sigma != 0. or self.consume_rng_when_deterministic:
            noise = torch.randn(x.shape, device=x.device) * temperature
            if sigma != 0.:
                noise = _from_rank0(unet, noise)
        x_prev, pred_x0 = ops.cfg_ddim_step(eps.float().contiguous(), x, self.ddim_coef, index, cfg=cfg, scale=scale,
                                            cond_f=cond_f, noise=noise if sigma != 0. else None)
        return x_prev, pred_x0


    # ---- the captured step -----------------------------------------------------------------------------------------
    def _graph_step(self, unet, x, c, uc, index, x0_emb, cond_frames, scale):
        """One p_sample_ddim as a single hipGraph replay (seer_ddim_step_begin -> the UNet's kernels -> seer_cfg_ddim_step_dev):
        no torch kernel between two UNet evaluations except the reference's per-step RNG draw, no host-written scalars -- the
        schedule index lives in device memory and the captured step counts it down itself.  Returns None when this step cannot
        take the path (sharded model, capture refused): the caller then runs the launches one by one."""
        from .unet import SeerUNet
        if not isinstance(unet, SeerUNet) or unet._shard is not None or unet._ops_backend is not ops \
                or unet.config.center_input_sample or getattr(self, "_step_graph_broken", False):
            return None
        if unet._engine is None or unet._engine.device != x.device:
            unet.prepare()
        eng = unet._engine
        if getattr(eng, "_graph_broken", False):
            return None
        cfg = uc is not None
        if cfg:
            cached = getattr(self, "_cfg_inputs", None)
            if cached is None or cached[0] is not c or cached[1] is not uc or cached[3] != (c._version, uc._version):
                cached = (c, uc, torch.cat([uc, c]).contiguous(), (c._version, uc._version))
                self._cfg_inputs = cached
            context = cached[2]
        else:
            context = c
        b, Cc, Fp, h, w = x.shape
        f1 = 0 if x0_emb is None else x0_emb.shape[2]
        reps = 2 if cfg else 1
        if context.dim() == 3:
            context = context[:, None].expand(-1, f1 + Fp, -1, -1)
        if context.shape[0] != reps * b or context.shape[1] != f1 + Fp or h % 8 or w % 8:
            return None
        ctx, L = eng._context(context)          # new prompt: the static context / K|V buffers are refreshed in place
        # (the guidance scale is only part of a CFG step: a plain step at another scale is the same graph)
        key = ("step", (b, Cc, Fp, h, w), f1, cfg, int(cond_frames), float(scale) if cfg else None, L, tuple(ctx.shape))
        G = eng.graph_get(key)
        if G is None:
            G = self._capture_step(eng, key, x, x0_emb, reps, cfg, int(cond_frames), float(scale), ctx, L)
            if G is None:
                return None
        # inputs that are read by address: refreshed in place when the caller brings new ones (once per sample / per schedule)
        # (the keyed tensors are kept alive with their keys: a freed tensor's address handed to the next sample's tensor must not
        #  look like "unchanged")
        if x0_emb is not None:
            k0 = (x0_emb.data_ptr(), x0_emb._version)
            if G["x0_key"] != k0 or G["x0_ref"] is not x0_emb:
                G["x0"].copy_(x0_emb)
                G["x0_key"], G["x0_ref"] = k0, x0_emb
        ksch = (self.ddim_coef.data_ptr(), self._t_table.data_ptr(), self.ddim_coef.shape[0])
        if G["sched_key"] != ksch or G["sched_ref"][0] is not self.ddim_coef:
            n = self.ddim_coef.shape[0]
            if n > G["coef"].shape[0]:
                return None                         # a longer schedule than the one captured for: take the eager path
            G["coef"][:n].copy_(self.ddim_coef)
            G["ttab"][:n].copy_(self._t_table)
            G["sched_key"], G["sched_ref"] = ksch, (self.ddim_coef, self._t_table)
            G["expect"] = None
        if x is not G["x"] and x.data_ptr() != G["x"].data_ptr():
            G["x"].copy_(x)
        if G["expect"] != index:
            G["step"][:1].fill_(int(index))         # start of a chain (or a caller that jumps): the only host-written index
        if self.consume_rng_when_deterministic:
            torch.randn(x.shape, device=x.device)   # ddim_video.py:234 draws every step, also at sigma = 0: keep the RNG stream
        G["graph"].replay()
        G["expect"] = index - 1
        if self.static_step_outputs:
            return G["x"], G["pred"]
        return G["x"].clone(), G["pred"].clone()

    def _capture_step(self, eng, key, x, x0_emb, reps, cfg, cond_frames, scale, ctx, L):
        dev = x.device
        b, Cc, Fp, h, w = x.shape
        f1 = 0 if x0_emb is None else x0_emb.shape[2]
        nsched = max(int(self.ddim_coef.shape[0]), 64)
        G = dict(x=torch.empty_like(x), pred=torch.empty_like(x),
                 x0=(torch.empty_like(x0_emb, dtype=torch.float32).contiguous() if x0_emb is not None else None), x0_key=None,
                 sample=torch.empty((reps * b, Cc, f1 + Fp, h, w), device=dev, dtype=torch.float32),
                 t=torch.empty((reps * b,), device=dev, dtype=torch.long), step=torch.zeros((2,), device=dev, dtype=torch.int32),
                 coef=torch.zeros((nsched, 4), device=dev, dtype=torch.float32),
                 ttab=torch.zeros((nsched,), device=dev, dtype=torch.long), sched_key=None, sched_ref=(None, None), x0_ref=None,
                 expect=None)
        G["coef"][:, 0] = 1.0                        # a_t = 1 in the unused rows: the warm-up below must stay finite
        G["x"].copy_(x)
        if x0_emb is not None:
            G["x0"].copy_(x0_emb)

        def body():
            ops.ddim_step_begin(G["x0"], G["x"], G["ttab"], G["step"], reps, G["sample"], G["t"])
            eps = eng._forward(G["sample"], G["t"], ctx, L, cond_frames)
            ops.cfg_ddim_step_dev(eps, G["x"], G["coef"], G["step"], cfg=cfg, scale=scale, cond_f=f1, x_prev=G["x"],
                                  pred_x0=G["pred"])
        try:
            body()                                   # warm-up: K|V / rotary caches, allocations
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
        except Exception as e:      # noqa: BLE001  capture refused: keep the launch-by-launch path
            self._step_graph_broken = True
            import warnings
            warnings.warn(f"hipGraph capture of the sampler step failed ({type(e).__name__}: {e}); stepping launch by launch")
            return None
        G["graph"] = g
        eng.graph_put(key, G)
        return G


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
