"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REAL reference modules (imported from
/root/reference through oracle/ref_import.py) on CPU fp32.  Run in the build container:

    python -m oracle.make_goldens

Weights come from seervideoldm_amd.synth (closed form, not stored); inputs and the reference's outputs are stored.
Every fixture is data: inputs + expected outputs.  Nothing of the reference's source is written anywhere.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import ref_import  # noqa: E402
from seervideoldm_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden"

TINY_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(32, 64, 64, 64),
                 cross_attention_dim=64, attention_head_dim=8, layers_per_block=2)
TINY_VAE = dict(ch=32, ch_mult=(1, 2, 2, 2), num_res_blocks=1, z_channels=4, out_ch=3)


def _save(name, **arrs):
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / name, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                       for k, v in arrs.items()})
    print(f"wrote {name}: " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrs.items()))


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _load_synth(module, prefix=""):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.synth_state_dict({prefix + k: s for k, s in shapes.items()})
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()}, strict=True)
    return sd


@torch.no_grad()
def main():
    ref = ref_import.load_reference()

    # ---- 1. schedule tables (ddim_video.py:27-68)
    for S in (4, 30, 50):
        smp = ref.ddim.DDIMSampler("cpu")
        smp.make_schedule(ddim_num_steps=S, ddim_eta=0.0, verbose=False)
        n = len(smp.ddim_timesteps)
        _save(f"schedule_S{S}.npz", ddim_timesteps=np.asarray(smp.ddim_timesteps),
              alphas=np.asarray([float(smp.ddim_alphas[i]) for i in range(n)]),
              alphas_prev=np.asarray([float(smp.ddim_alphas_prev[i]) for i in range(n)]),
              sigmas=np.asarray([float(smp.ddim_sigmas[i]) for i in range(n)]),
              sqrt_one_minus_alphas=np.asarray([float(smp.ddim_sqrt_one_minus_alphas[i]) for i in range(n)]),
              betas_0_999=np.asarray([float(smp.betas[0]), float(smp.betas[999])]),
              alphas_cumprod_0_999=np.asarray([float(smp.alphas_cumprod[0]), float(smp.alphas_cumprod[999])]))

    # ---- 2. per-op: ResnetBlock3D (resnet.py:106-208)
    for tag, cin, cout in (("same", 64, 64), ("widen", 96, 64)):
        blk = ref.resnet.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=128, eps=1e-5, groups=32,
                                       non_linearity="silu").eval()
        _load_synth(blk, f"resnet_{tag}.")
        x, temb = _randn((2, cin, 3, 8, 8), 1), _randn((2, 128), 2)
        _save(f"op_resnet_{tag}.npz", x=x, temb=temb, y=blk(x, temb))

    # ---- per-op: SpatialTransformer3D text block (attention.py:97-145,265-327)
    st = ref.attention.SpatialTransformer3D(64, 8, 8, depth=1, context_dim=48, text_frame_condition=True).eval()
    ref_import.enable_xformers_path(st)
    _load_synth(st, "st_text.")
    x, ctx = _randn((2, 64, 3, 8, 8), 3), _randn((2, 3, 77, 48), 4)
    _save("op_transformer_text.npz", x=x, context=ctx, y=st(x, context=ctx))

    # ---- per-op: temporal SpatialTransformer3D in the three window regimes x cond_frame (attention.py:181-248,632-703)
    for H, C, Fr in ((32, 64, 2), (16, 64, 3), (8, 64, 4), (4, 64, 5)):
        tt = ref.attention.SpatialTransformer3D(C, 8, C // 8, depth=1, context_dim=None, temporal=True, causal=True).eval()
        ref_import.enable_xformers_path(tt)
        _load_synth(tt, "st_temporal.")
        x = _randn((1 if H >= 16 else 2, C, Fr, H, H), 10 + H)
        ys = {f"y_cond{cf}": tt(x, cond_frame=cf) for cf in (0, 2 if Fr > 2 else 1)}
        _save(f"op_transformer_temporal_H{H}.npz", x=x, **ys)

    # ---- per-op: Downsample3D / Upsample3D (resnet.py:18-104)
    dn = ref.resnet.Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op").eval()
    _load_synth(dn, "down.")
    up = ref.resnet.Upsample3D(64, use_conv=True, out_channels=64).eval()
    _load_synth(up, "up.")
    x = _randn((2, 64, 2, 8, 8), 20)
    _save("op_updown.npz", x=x, y_down=dn(x), y_up=up(x))

    # ---- 3. tiny SeerUNet end to end (unet_3d_condition.py:283-376)
    unet = ref.unet.SeerUNet(**TINY_UNET).eval()
    ref_import.enable_xformers_path(unet)
    shapes = synth.unet_param_shapes(TINY_UNET)
    assert set(shapes) == set(unet.state_dict().keys())
    unet.load_state_dict(synth.synth_state_dict(shapes), strict=True)
    x, ctx = _randn((2, 4, 4, 16, 16), 30), _randn((2, 4, 77, 64), 31)
    t = torch.tensor([501, 501])
    _save("unet_tiny.npz", sample=x, timestep=t, context=ctx, y_cond0=unet(x, t, ctx, cond_frame=0),
          y_cond2=unet(x, t, ctx, 2), y_scalar_t=unet(x, 7, ctx))

    # ---- 4. VAE decoder (vendored twin, model.py:462-568) + post_quant_conv
    dec = ref.vae.Decoder(ch=TINY_VAE["ch"], out_ch=3, ch_mult=TINY_VAE["ch_mult"], num_res_blocks=TINY_VAE["num_res_blocks"],
                          attn_resolutions=[], in_channels=3, resolution=64, z_channels=4).eval()
    vshapes = synth.vae_param_shapes(**TINY_VAE)
    vsd = synth.synth_state_dict(vshapes)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in vsd.items() if k.startswith("decoder.")}, strict=True)
    pq = torch.nn.Conv2d(4, 4, 1)
    pq.load_state_dict({"weight": vsd["post_quant_conv.weight"], "bias": vsd["post_quant_conv.bias"]})
    z = _randn((3, 4, 8, 8), 40)
    _save("vae_tiny.npz", z=z, y=dec(pq(z)))

    # ---- 5. sampler: one p_sample_ddim step with CFG and a full 4-step ddim_sample incl. decode
    class _Vae:   # diffusers AutoencoderKL.decode(z).sample surface over the vendored decoder
        def decode(self, zz):
            import types
            return types.SimpleNamespace(sample=dec(pq(zz)))

    smp = ref.ddim.DDIMSampler("cpu")
    b, f1, Fp = 1, 2, 2
    x0_emb = _randn((b, 4, f1, 16, 16), 50) * 0.9
    c = _randn((b, f1 + Fp, 77, 64), 51)
    uc = _randn((b, 1, 77, 64), 52).expand(-1, f1 + Fp, -1, -1).contiguous()
    noise = _randn((b, 4, Fp, 16, 16), 53)
    smp.make_schedule(ddim_num_steps=4, ddim_eta=0.0, verbose=False)
    ts = torch.full((b,), 751, dtype=torch.long)
    torch.manual_seed(123)
    x_prev, pred_x0 = smp.p_sample_ddim(unet, noise, c, ts, index=3, is_3d=True, x0_emb=x0_emb,
                                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc)
    _save("ddim_step.npz", x=noise, x0_emb=x0_emb, c=c, uc=uc, x_prev=x_prev, pred_x0=pred_x0)
    torch.manual_seed(123)
    clip = ref.glue.ddim_sample(smp, unet, _Vae(), shape=(b, 4, Fp, 16, 16), c=c, start_code=noise, x0_emb=x0_emb,
                                ddim_steps=4, scale=7.5, uc=uc)
    torch.manual_seed(123)
    lat, _ = smp.sample(unet=unet, S=4, conditioning=c, batch_size=b, shape=(4, Fp, 16, 16), x0_emb=x0_emb,
                        verbose=False, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0,
                        x_T=noise, is_3d=True)
    _save("ddim_sample_tiny.npz", start_code=noise, x0_emb=x0_emb, c=c, uc=uc, latent=lat, clip=clip)

    gen_fstext(ref)
    gen_vae_encode(ref)
    with torch.enable_grad():
        gen_train(ref)


TINY_FSTEXT = dict(num_frames=6, num_layers=2, channels=192, n_heads=2, cross_attention_dim=192)


@torch.no_grad()
def gen_fstext(ref):
    """6. FSTextTransformer (unet_3d_condition.py:379-484), the step before the path: head dim 96 like the real model
    (768 / 8), two layers, evaluated at its native frame count and after set_numframe(4) (nearest resize of pos_embed)."""
    c = TINY_FSTEXT
    m = ref.unet.FSTextTransformer(num_frames=c["num_frames"], in_channels=c["channels"], out_channels=c["channels"],
                                   n_heads=c["n_heads"], num_layers=c["num_layers"],
                                   cross_attention_dim=c["cross_attention_dim"]).eval()
    ref_import.enable_xformers_path(m)
    shapes = synth.fstext_param_shapes(**c)
    assert set(shapes) == set(m.state_dict().keys())
    m.load_state_dict(synth.synth_state_dict(shapes), strict=True)
    ctx = _randn((2, 77, c["channels"]), 60)
    out = {}
    for Fr in (6, 4):
        m.set_numframe(Fr)
        out[f"y_F{Fr}"] = m(context=ctx)
    _save("fstext_tiny.npz", context=ctx, **out)


@torch.no_grad()
def gen_vae_encode(ref):
    """7. VAE encoder (vendored twin, ldm/modules/diffusionmodules/model.py:368-460) + quant_conv + the reference's
    DiagonalGaussianDistribution.sample (ldm/modules/distributions/distributions.py:24-37) under a fixed seed."""
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    enc = ref.vae.Encoder(ch=TINY_VAE["ch"], out_ch=3, ch_mult=TINY_VAE["ch_mult"], num_res_blocks=TINY_VAE["num_res_blocks"],
                          attn_resolutions=[], in_channels=3, resolution=64, z_channels=4, double_z=True).eval()
    shapes = synth.vae_encoder_param_shapes(ch=TINY_VAE["ch"], ch_mult=TINY_VAE["ch_mult"],
                                            num_res_blocks=TINY_VAE["num_res_blocks"], z_channels=4)
    sd = synth.synth_state_dict(shapes)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True)
    qc = torch.nn.Conv2d(8, 8, 1)
    qc.load_state_dict({"weight": sd["quant_conv.weight"], "bias": sd["quant_conv.bias"]})
    x = _randn((2, 3, 64, 64), 70)
    moments = qc(enc(x))
    torch.manual_seed(71)
    sample = DiagonalGaussianDistribution(moments).sample()
    torch.manual_seed(71)
    noise = torch.randn(sample.shape)
    _save("vae_enc_tiny.npz", x=x, moments=moments, noise=noise, sample=sample)


TINY_TRAIN_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(32, 64, 64, 64),
                       cross_attention_dim=64, attention_head_dim=8, layers_per_block=2)
TINY_TRAIN_FSTEXT = dict(num_frames=16, num_layers=1, channels=64, n_heads=2, cross_attention_dim=64)
TRAIN_HP = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=0.3)
TRAIN_FULL_KEYS_U = ("down_blocks.0.temporal_attentions.0.transformer_blocks.0.attn1.to_q.weight",
                     "down_blocks.0.temporal_attentions.0.norm.weight",
                     "mid_block.temporal_attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
                     "up_blocks.3.temporal_attentions.1.proj_out.bias",
                     "up_blocks.1.temporal_attentions.0.transformer_blocks.0.norm3.weight")
TRAIN_FULL_KEYS_F = ("learnable_query", "trf_blocks.0.transformer_blocks.0.attn2.to_k.weight",
                     "trf_blocks.0.transformer_blocks.1.attn1.to_q.weight", "trf_blocks.0.transformer_blocks.1.ff.net.2.bias",
                     "norm.weight")


def gen_train(ref):
    """8. one fine-tuning step of train.py:319-389 on the real reference modules: FSText -> SeerUNet -> eps-MSE -> backward ->
    clip_grad_norm_(sunet) -> AdamW, at 4 frames with 2 conditioning frames.  Stored: the inputs, the loss, per-parameter
    gradient statistics (L2 norm, sum) for EVERY trainable tensor, full gradients and updated values of a few tensors."""
    unet = ref.unet.SeerUNet(**TINY_TRAIN_UNET)
    ref_import.enable_xformers_path(unet)
    ushapes = synth.unet_param_shapes(TINY_TRAIN_UNET)
    unet.load_state_dict(synth.synth_state_dict(ushapes), strict=True)
    c = TINY_TRAIN_FSTEXT
    fst = ref.unet.FSTextTransformer(num_frames=c["num_frames"], in_channels=c["channels"], out_channels=c["channels"],
                                     n_heads=c["n_heads"], num_layers=c["num_layers"], cross_attention_dim=c["cross_attention_dim"])
    ref_import.enable_xformers_path(fst)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**c)), strict=True)
    Fr, cond = 4, 2
    fst.set_numframe(Fr)                                                  # train.py:187
    unet.requires_grad_(False)                                            # train.py:188-192
    for name, module in unet.named_modules():
        if name.endswith(("temporal_attentions",)):
            for prm in module.parameters():
                prm.requires_grad = True
    unet.train(); fst.train()
    params = [p for p in unet.parameters() if p.requires_grad] + list(fst.parameters())
    hp = TRAIN_HP
    opt = torch.optim.AdamW(params, lr=hp["lr"], betas=hp["betas"], weight_decay=hp["weight_decay"], eps=hp["eps"])
    latents_x0, latents = _randn((1, 4, cond, 16, 16), 80), _randn((1, 4, Fr - cond, 16, 16), 81)
    noise, text = _randn((1, 4, Fr - cond, 16, 16), 82), _randn((1, 77, 64), 83)
    t = torch.tensor([417])
    # DDPMScheduler.add_noise with the SD scaled-linear schedule (train.py:234,363)
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, 0)
    a = acp[t].reshape(-1, 1, 1, 1, 1)
    noisy = a.sqrt() * latents + (1 - a).sqrt() * noise
    x = torch.cat([latents_x0, noisy], 2)
    text_seq = fst(context=text)
    pred = unet(x, t, text_seq, cond)
    loss = torch.nn.functional.mse_loss(pred[:, :, cond:], noise, reduction="none").mean([1, 2, 3, 4]).mean()
    # the same step with `text_loss: True` (train.py:346-347,377-378): only the FSTextTransformer gradients change
    loss_text = torch.nn.functional.mse_loss(text_seq.mean(1), text.clone().detach(), reduction="none").mean([1, 2]).mean()
    (loss + loss_text).backward(retain_graph=True)
    tl_stats = np.asarray([[float(p.grad.norm()), float(p.grad.sum())] for p in fst.parameters()])
    tl_q = fst.trf_blocks[0].transformer_blocks[1].attn1.to_q.weight.grad.detach().clone()
    for p in list(unet.parameters()) + list(fst.parameters()):
        p.grad = None
    loss.backward()
    un = {k: p for k, p in unet.named_parameters() if p.requires_grad}
    fn = dict(fst.named_parameters())
    stats = lambda d: np.asarray([[float(p.grad.norm()) if p.grad is not None else 0.0,
                                   float(p.grad.sum()) if p.grad is not None else 0.0] for p in d.values()])
    out = dict(latents_x0=latents_x0, latents=latents, noise=noise, text=text, timestep=t, alphas_cumprod=acp, model_input=x,
               loss=loss.detach(), pred=pred.detach(), unet_keys=np.asarray(list(un)), fstext_keys=np.asarray(list(fn)),
               unet_grad_stats=stats(un), fstext_grad_stats=stats(fn))
    out.update(loss_text=loss_text.detach(), fstext_grad_stats_text_loss=tl_stats,
               **{"gft:trf_blocks.0.transformer_blocks.1.attn1.to_q.weight": tl_q})
    for k in TRAIN_FULL_KEYS_U:
        out["gu:" + k] = un[k].grad.detach().clone()
    for k in TRAIN_FULL_KEYS_F:
        out["gf:" + k] = fn[k].grad.detach().clone()
    total = torch.nn.utils.clip_grad_norm_(unet.parameters(), hp["max_grad_norm"])      # train.py:384
    opt.step()
    out["unet_grad_norm"] = total
    for k in TRAIN_FULL_KEYS_U:
        out["pu:" + k] = un[k].detach().clone()
    for k in TRAIN_FULL_KEYS_F:
        out["pf:" + k] = fn[k].detach().clone()
    _save("train_tiny.npz", **out)


if __name__ == "__main__":
    if "train" in sys.argv[1:]:
        gen_train(ref_import.load_reference())
    elif "fstext" in sys.argv[1:]:
        gen_fstext(ref_import.load_reference())
    elif "vae_encode" in sys.argv[1:]:
        gen_vae_encode(ref_import.load_reference())
    else:
        main()
