"""VAE encode of the conditioning frames (the step before the path, inference_img.py:166-170).

CPU: the oracle restatement against the reference's own Encoder + quant_conv + DiagonalGaussianDistribution.sample
(tests/golden/vae_enc_tiny.npz, made by oracle/make_goldens.py from ldm/modules/diffusionmodules/model.py), key-layout
conversion.  GPU: the HIP path against the oracle."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import seer_oracle as O
from seervideoldm_amd import AutoencoderKL, synth
from seervideoldm_amd.vae import ldm_to_diffusers_vae, vae_decoder_shapes, vae_encoder_shapes

G = Path(__file__).parent / "golden"
TINY = dict(ch=32, ch_mult=(1, 2, 2, 2), num_res_blocks=1, z_channels=4)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def test_oracle_matches_reference_golden():
    g = {k: torch.from_numpy(v) for k, v in np.load(G / "vae_enc_tiny.npz").items()}
    sd = synth.synth_state_dict(synth.vae_encoder_param_shapes(**TINY))
    m = O.vae_encode_moments(sd, g["x"], ch_mult=TINY["ch_mult"], num_res_blocks=TINY["num_res_blocks"])
    torch.testing.assert_close(m, g["moments"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(O.gaussian_sample(m, g["noise"]), g["sample"], rtol=1e-4, atol=2e-5)


def test_key_layout_conversion_and_partial_checkpoints():
    """ldm -> diffusers names for both halves; a checkpoint with one half loads strictly, the other half stays unusable"""
    enc = synth.synth_state_dict(synth.vae_encoder_param_shapes())
    dec = synth.synth_state_dict(synth.vae_param_shapes())
    conv = ldm_to_diffusers_vae({**enc, **dec}, 4)
    want = {**vae_encoder_shapes(), **vae_decoder_shapes()}
    assert set(conv) == set(want)
    assert all(tuple(conv[k].shape) == tuple(want[k]) for k in want)
    assert sum(v.numel() for k, v in conv.items() if k.startswith(("encoder.", "quant_conv."))) == 34_163_664   # SD VAE encoder
    vae = AutoencoderKL()
    vae.load_state_dict(ldm_to_diffusers_vae(dec, 4), strict=True)
    assert vae._loaded == {"encoder": False, "decoder": True}
    with pytest.raises(RuntimeError):
        half = dict(list(ldm_to_diffusers_vae(enc, 4).items())[:-3])
        AutoencoderKL().load_state_dict(half, strict=True)
    vae.load_state_dict(ldm_to_diffusers_vae(enc, 4), strict=True)
    assert vae._loaded == {"encoder": True, "decoder": True}


def test_from_pretrained_layout(tmp_path):
    """`AutoencoderKL.from_pretrained(model, subfolder="vae")` (inference_img.py:69): config.json + weights by name"""
    import json
    cfgd = dict(_class_name="AutoencoderKL", in_channels=3, out_channels=3, block_out_channels=[32, 64], layers_per_block=1,
                latent_channels=4, norm_num_groups=8, sample_size=64, act_fn="silu")
    src = AutoencoderKL(block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in src.state_dict().items()})
    (tmp_path / "vae").mkdir()
    (tmp_path / "vae" / "config.json").write_text(json.dumps(cfgd))
    torch.save(sd, tmp_path / "vae" / "diffusion_pytorch_model.bin")
    vae = AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    assert vae.config.block_out_channels == (32, 64) and vae._loaded == {"encoder": True, "decoder": True}
    for k, v in vae.state_dict().items():
        assert torch.equal(v, sd[k]), k


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(2, 64, 64), (1, 64, 128), (1, 64, 96), (2, 32, 96)])   # the last two: 96 and 48 mid-attention tokens (zero-padded keys)
def test_hip_encode_matches_oracle(N, H, W):
    dev = torch.device("cuda:0")
    kw = dict(ch=128, ch_mult=(1, 1, 2, 2), num_res_blocks=1, z_channels=4)       # >= 4 channels per GroupNorm group
    sd = synth.synth_state_dict(synth.vae_encoder_param_shapes(**kw))
    vae = AutoencoderKL(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    vae.load_state_dict(ldm_to_diffusers_vae(sd, 4), strict=True)
    vae = vae.to(dev)
    x = _randn((N, 3, H, W), 5).clamp(-1, 1)
    ref = O.vae_encode_moments(sd, x, ch_mult=kw["ch_mult"], num_res_blocks=1)
    dist = vae.encode(x.to(dev)).latent_dist
    got = dist.parameters.cpu()
    assert got.shape == ref.shape == (N, 8, H // 8, W // 8)
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel                       # bf16 activations / weights vs the fp32 reference arithmetic
    # sampling: same generator -> same noise as torch.randn on the device; mean + std * noise exactly
    gen = torch.Generator(device=dev).manual_seed(9)
    s = dist.sample(generator=gen)
    noise = torch.randn(s.shape, generator=torch.Generator(device=dev).manual_seed(9), device=dev)
    torch.testing.assert_close(s.cpu(), O.gaussian_sample(got, noise.cpu()), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dist.mode().cpu(), got[:, :4], rtol=0, atol=0)
    with pytest.raises(RuntimeError):
        vae.decode(torch.zeros(1, 4, 8, 8, device=dev))      # no decoder weights in this checkpoint


@pytest.mark.gpu
def test_hip_encode_full_size():
    """SD-VAE encoder (34.2 M params) on two 256x256 conditioning frames, as inference_img.py:166-170 does"""
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(synth.vae_encoder_param_shapes())
    vae = AutoencoderKL()
    vae.load_state_dict(ldm_to_diffusers_vae(sd, 4), strict=True)
    vae = vae.to(dev)
    x = torch.tanh(_randn((2, 3, 256, 256), 3))
    m1 = vae.encode(x.to(dev)).latent_dist.parameters
    m2 = vae.encode(x.to(dev)).latent_dist.parameters
    assert m1.shape == (2, 8, 32, 32) and torch.equal(m1, m2)
    ref = O.vae_encode_moments(sd, x)
    rel = ((m1.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel
