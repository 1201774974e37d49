"""Object-lifetime tests on a real MI355X: every module caches packed weights, tables, K/V projections or captured graphs.
A long-lived object driven through CHANGING shapes / prompts / weights must give, bit for bit, what a fresh object gives."""
import pytest
import torch

from seervideoldm_amd import AutoencoderKL, FSTextTransformer, SeerUNet, synth
from seervideoldm_amd.trainer import SeerTrainer
from seervideoldm_amd.vae import ldm_to_diffusers_vae

pytestmark = pytest.mark.gpu

UNET = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
FST = dict(num_frames=16, num_layers=1, channels=192, n_heads=2, cross_attention_dim=192)
VAE = dict(ch=128, ch_mult=(1, 1, 2, 2), num_res_blocks=1)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _unet(dev):
    m = SeerUNet(**UNET)
    m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(UNET)), strict=True)
    return m.to(dev).eval()


def _fst(dev):
    m = FSTextTransformer(num_frames=16, in_channels=192, out_channels=192, n_heads=2, num_layers=1, cross_attention_dim=192)
    m.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**FST)), strict=True)
    return m.to(dev).eval()


def _vae(dev):
    vsd = {**synth.synth_state_dict(synth.vae_param_shapes(**VAE)),
           **synth.synth_state_dict(synth.vae_encoder_param_shapes(**VAE, z_channels=4))}
    v = AutoencoderKL(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    v.load_state_dict(ldm_to_diffusers_vae(vsd, 4), strict=True)
    return v.to(dev)


def test_unet_changing_shapes_under_graph_replay(device):
    """seven (batch, frames, size, cond_frame) combinations round-robin, twice, with use_graph (the graph table holds 4)"""
    cases = [(1, 2, 8, 0), (2, 2, 8, 0), (1, 3, 8, 1), (1, 2, 16, 0), (2, 3, 8, 2), (1, 4, 8, 0), (1, 2, 8, 1)]
    inputs = [(_randn((b, 4, f, h, h), 10 + i).to(device), torch.tensor([100 + i] * b, device=device),
               _randn((b, f, 77, 192), 50 + i).to(device), cf) for i, (b, f, h, cf) in enumerate(cases)]
    ref = []
    for x, t, c, cf in inputs:
        ref.append(_unet(device)(x, t, c, cond_frame=cf).clone())          # a fresh object per case, eager
    m = _unet(device)
    m.use_graph = True
    for rnd in range(2):
        for i, (x, t, c, cf) in enumerate(inputs):
            assert torch.equal(m(x, t, c, cond_frame=cf), ref[i]), (rnd, i)


def test_unet_graph_replay_between_eager_forwards_of_two_shapes(device):
    """The fixed-point accumulators of an evaluation (ops.FxArena) are zeroed by reset() at its head.  A replayed hipGraph adds into
    the slots baked into it without the host-side bump allocator seeing it: capture the LARGE shape, run a SMALL shape eagerly
    (fewer slots handed out), replay the large graph, then run the large shape EAGERLY on the same engine (what a return_attn
    forward or the capture-failure fallback does) -- its slots beyond the small shape's must have been zeroed too."""
    big = (_randn((2, 4, 3, 16, 16), 1).to(device), torch.tensor([300, 300], device=device), _randn((2, 3, 77, 192), 2).to(device))
    small = (_randn((1, 4, 2, 8, 8), 3).to(device), torch.tensor([301], device=device), _randn((1, 2, 77, 192), 4).to(device))
    ref_big, ref_small = _unet(device)(*big, cond_frame=0).clone(), _unet(device)(*small, cond_frame=0).clone()
    m = _unet(device)
    m.use_graph = True
    assert torch.equal(m(*big, cond_frame=0), ref_big)            # eager warm-up + capture + first replay
    m.use_graph = False
    assert torch.equal(m(*small, cond_frame=0), ref_small)        # eager, fewer accumulator slots
    m.use_graph = True
    assert torch.equal(m(*big, cond_frame=0), ref_big)            # replay: adds into the large shape's slots
    m.use_graph = False
    assert torch.equal(m(*big, cond_frame=0), ref_big)            # eager on the same arena
    out, _attn = m(*big, cond_frame=0, return_attn=True)          # the analysis path is eager too
    assert torch.equal(out, ref_big)


def test_fstext_changing_frames_and_batch(device):
    ctxs = [_randn((b, 77, 192), 20 + b).to(device) for b in (1, 2, 3)]
    plan = [(4, 0), (6, 1), (16, 2), (4, 1), (6, 0), (4, 0)]
    m = _fst(device)
    for Fr, ci in plan:
        m.set_numframe(Fr)
        got = m(context=ctxs[ci])
        fresh = _fst(device)
        fresh.set_numframe(Fr)
        assert torch.equal(got, fresh(context=ctxs[ci])), (Fr, ci)


def test_vae_changing_batch_and_resolution(device):
    v = _vae(device)
    for i, (n, h) in enumerate([(4, 16), (1, 8), (6, 16), (2, 32), (4, 16)]):
        z = _randn((n, 4, h, h), 30 + i).to(device)
        img = v.decode(z).sample
        fresh = _vae(device)
        assert torch.equal(img, fresh.decode(z).sample), ("decode", n, h)
        x = torch.tanh(_randn((n, 3, 8 * h, 8 * h), 40 + i)).to(device)
        assert torch.equal(v.encode(x).latent_dist.mode(), fresh.encode(x).latent_dist.mode()), ("encode", n, h)


def test_trainer_changing_shapes_under_graph_replay_and_sampling_after_training(device):
    unet, fst = _unet(device), _fst(device)
    tr = SeerTrainer(unet, fst, lr=1e-4, max_grad_norm=0.3)
    cases = [(1, 3, 8, 1), (1, 4, 8, 2), (2, 3, 8, 1), (1, 3, 8, 1), (1, 4, 8, 2)]
    ref_unet, ref_fst = _unet(device), _fst(device)
    ref = SeerTrainer(ref_unet, ref_fst, lr=1e-4, max_grad_norm=0.3)                   # the same steps, eager
    for i, (b, f, h, cf) in enumerate(cases):
        x, noise = _randn((b, 4, f, h, h), 60 + i).to(device), _randn((b, 4, f - cf, h, h), 70 + i).to(device)
        text, t = _randn((b, 77, 192), 80 + i).to(device), torch.tensor([200 + i] * b, device=device)
        fst.set_numframe(f)
        ref_fst.set_numframe(f)
        la = tr.forward_backward(x, noise, t, text, cf, use_graph=True)
        lb = ref.forward_backward(x, noise, t, text, cf)
        assert float(la) == float(lb) and torch.equal(tr.pu.g, ref.pu.g) and torch.equal(tr.pf.g, ref.pf.g), i
        tr.optimizer_step()
        ref.optimizer_step()
    assert torch.equal(tr.pu.p, ref.pu.p) and torch.equal(tr.pf.p, ref.pf.p)
    # the trained objects sample with the trained weights once the masters are pushed back (and only then)
    x, t, c = _randn((1, 4, 3, 8, 8), 90).to(device), torch.tensor([50], device=device), _randn((1, 3, 77, 192), 91).to(device)
    before = unet(x, t, c, cond_frame=1).clone()
    tr.sync_modules()
    after = unet(x, t, c, cond_frame=1).clone()
    assert not torch.equal(before, after)
    fresh = SeerUNet(**UNET)
    fresh.load_state_dict({k: v.cpu() for k, v in unet.state_dict().items()}, strict=True)
    assert torch.equal(after, fresh.to(device).eval()(x, t, c, cond_frame=1))
