"""End-to-end caller harness (inference_img.py:164-187): conditioning image + CLIP embeddings ->
VAE encode -> FSTextTransformer -> 4-step CFG DDIM -> VAE decode, the HIP pipeline against the same chain of oracle
functions fed with the same random draws (CPU generator for the start code, device generator for the latent sample)."""
import pytest
import torch

from oracle import seer_oracle as O
from seervideoldm_amd import AutoencoderKL, DDIMSampler, FSTextTransformer, SeerUNet, synth
from seervideoldm_amd.pipeline import generate_clips
from seervideoldm_amd.vae import ldm_to_diffusers_vae

UNET = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
FST = dict(num_frames=6, num_layers=2, channels=192, n_heads=2, cross_attention_dim=192)
VAE = dict(ch=128, ch_mult=(1, 1, 2, 2), num_res_blocks=1)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.gpu
def test_generate_clips_matches_oracle_chain():
    dev = torch.device("cuda:0")
    usd = synth.synth_state_dict(synth.unet_param_shapes(UNET))
    unet = SeerUNet(**UNET)
    unet.load_state_dict(usd, strict=True)
    fsd = synth.synth_state_dict(synth.fstext_param_shapes(**FST))
    fst = FSTextTransformer(num_frames=6, in_channels=192, out_channels=192, n_heads=2, num_layers=2, cross_attention_dim=192)
    fst.load_state_dict(fsd, strict=True)
    vsd = {**synth.synth_state_dict(synth.vae_param_shapes(**VAE)),
           **synth.synth_state_dict(synth.vae_encoder_param_shapes(**VAE, z_channels=4))}
    vae = AutoencoderKL(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    vae.load_state_dict(ldm_to_diffusers_vae(vsd, 4), strict=True)
    unet, fst, vae = unet.to(dev).eval(), fst.to(dev).eval(), vae.to(dev)

    b, f1, F_, R = 1, 1, 3, 128
    x0_image = torch.tanh(_randn((b, 3, 1, R, R), 1))
    text, empty = _randn((b, 77, 192), 2), _randn((b, 77, 192), 3)
    clips = generate_clips(unet, fst, vae, DDIMSampler(dev), x0_image.to(dev), text.to(dev), empty.to(dev), num_frames=F_,
                           cond_frames=f1, ddim_steps=4, scale=7.5, num_samples=2,
                           noise_generator=torch.Generator().manual_seed(11),
                           latent_generator=torch.Generator(device=dev).manual_seed(12))
    assert len(clips) == 2 and all(c.shape == (b, 3, F_ - f1, R, R) for c in clips)
    assert not torch.equal(clips[0], clips[1])            # the start code is redrawn for every sample (inference_img.py:187)

    # ---- the same chain on the CPU oracle with the same draws
    frames = x0_image.expand(-1, -1, f1, -1, -1).permute(0, 2, 1, 3, 4).reshape(b * f1, 3, R, R)
    mom = O.vae_encode_moments(vsd, frames, ch_mult=VAE["ch_mult"], num_res_blocks=1)
    lat_noise = torch.randn((b * f1, 4, R // 8, R // 8), generator=torch.Generator(device=dev).manual_seed(12), device=dev).cpu()
    lat = O.gaussian_sample(mom, lat_noise) * 0.18215
    x0_emb = lat.reshape(b, f1, 4, R // 8, R // 8).permute(0, 2, 1, 3, 4)
    c = O.fstext_forward(fsd, text, F_, heads=2)
    uc = empty.unsqueeze(1).expand(-1, F_, -1, -1)
    g = torch.Generator().manual_seed(11)
    unet_fn = lambda x, t, cc, cf: O.unet_forward(usd, UNET, x, t, cc, cond_frame=cf)
    for clip in clips:
        noise = torch.randn((b, 4, F_ - f1, R // 8, R // 8), generator=g)
        ref, _ = O.ddim_sample(unet_fn, vsd, (b, 4, F_ - f1, R // 8, R // 8), c, noise, x0_emb, ddim_steps=4, scale=7.5,
                               uc=uc, vae_kwargs=dict(ch_mult=VAE["ch_mult"], num_res_blocks=1))
        err = (clip.cpu() - ref).abs()
        print(f"[parity] pipeline clip: mean abs err {err.mean():.4g}, max {err.max():.4g}")
        # bf16 kernels through encode + text transformer + 4 CFG steps (scale 7.5) + decode vs fp32: stated tolerance
        assert err.mean() < 1.5e-2 and err.max() < 0.2


@pytest.mark.gpu
def test_prompt_after_prompt_with_graph_replay_equals_eager():
    """an eval-style loop (eval.py:174-231): one sampler, one set of models, a NEW prompt / conditioning image every iteration,
    old tensors dropped as the loop goes.  hipGraph replay must give the eager clips bit for bit for every prompt."""
    dev = torch.device("cuda:0")
    unet = SeerUNet(**UNET)
    unet.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(UNET)), strict=True)
    fst = FSTextTransformer(num_frames=6, in_channels=192, out_channels=192, n_heads=2, num_layers=2, cross_attention_dim=192)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**FST)), strict=True)
    vsd = {**synth.synth_state_dict(synth.vae_param_shapes(**VAE)),
           **synth.synth_state_dict(synth.vae_encoder_param_shapes(**VAE, z_channels=4))}
    vae = AutoencoderKL(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    vae.load_state_dict(ldm_to_diffusers_vae(vsd, 4), strict=True)
    unet, fst, vae = unet.to(dev).eval(), fst.to(dev).eval(), vae.to(dev)

    def loop(use_graph):
        unet.use_graph = use_graph
        unet._engine = None
        smp = DDIMSampler(dev)
        outs = []
        for i in range(4):
            img = torch.tanh(_randn((1, 3, 1, 64, 64), 100 + i)).to(dev)
            text, empty = _randn((1, 77, 192), 200 + i).to(dev), _randn((1, 77, 192), 300).to(dev)
            clip = generate_clips(unet, fst, vae, smp, img, text, empty, num_frames=3, cond_frames=1, ddim_steps=4, scale=7.5,
                                  noise_generator=torch.Generator().manual_seed(400 + i),
                                  latent_generator=torch.Generator(device=dev).manual_seed(500 + i))[0]
            outs.append(clip.clone())
            del img, text, empty, clip
        return outs
    eager, replay = loop(False), loop(True)
    unet.use_graph = False
    for i, (a, b) in enumerate(zip(eager, replay)):
        assert torch.equal(a, b), f"prompt {i}: max diff {(a - b).abs().max().item():.4g}"
    assert not torch.equal(eager[0], eager[1])
