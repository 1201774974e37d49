"""FSTextTransformer (the step before the path).

CPU: the oracle restatement against the reference's own module (tests/golden/fstext_tiny.npz, produced by
oracle/make_goldens.py from seer.models.unet_3d_condition.FSTextTransformer), and the product's kernel schedule through the
plain-torch stand-in of the kernel library.  GPU: the HIP path against the oracle, tiny and full size (768 channels,
8 heads -> head dim 96, 8 layers, F = 12 resized from the checkpoint's 16 frames)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import seer_oracle as O
from seervideoldm_amd import FSTextTransformer, synth
from tests import torch_ops_backend as tob

G = Path(__file__).parent / "golden"
TINY = dict(num_frames=6, num_layers=2, channels=192, n_heads=2, cross_attention_dim=192)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _module(cfg, device="cpu"):
    sd = synth.synth_state_dict(synth.fstext_param_shapes(**cfg))
    m = FSTextTransformer(num_frames=cfg["num_frames"], in_channels=cfg["channels"], out_channels=cfg["channels"],
                          n_heads=cfg["n_heads"], num_layers=cfg["num_layers"], cross_attention_dim=cfg["cross_attention_dim"])
    m.load_state_dict(sd, strict=True)
    return sd, m.to(device)


def test_state_dict_layout():
    """same keys / shapes as a Seer `pytorch_model_1.bin` (inference_img.py:80,100-101): 182.6 M parameters"""
    shapes = synth.fstext_param_shapes()
    n = sum(int(np.prod(s)) for k, s in shapes.items() if not k.endswith("freqs"))
    assert abs(n / 1e6 - 182.645) < 0.01
    m = FSTextTransformer(num_frames=16, num_layers=1)
    keys = set(m.state_dict().keys())
    assert {"learnable_query", "pos_embed", "norm.weight", "trf_blocks.0.transformer_blocks.0.attn2.to_k.weight",
            "trf_blocks.0.transformer_blocks.1.attn1.rotary_emb.freqs"} <= keys
    assert not any(".transformer_blocks.1.attn2." in k or ".transformer_blocks.1.norm2." in k for k in keys)
    with pytest.raises(NotImplementedError):
        FSTextTransformer(num_frames=4, in_channels=1024, out_channels=768)


@pytest.mark.parametrize("Fr", [6, 4])
def test_oracle_matches_reference_golden(Fr):
    g = {k: torch.from_numpy(v) for k, v in np.load(G / "fstext_tiny.npz").items()}
    sd = synth.synth_state_dict(synth.fstext_param_shapes(**TINY))
    y = O.fstext_forward(sd, g["context"], Fr, heads=TINY["n_heads"])
    torch.testing.assert_close(y, g[f"y_F{Fr}"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("Fr,b", [(6, 1), (4, 2)])
def test_schedule_matches_oracle_cpu(Fr, b):
    """host logic (weight packing, fused q|k|v, strided frame sequences, rotary rows, pos_embed resize) on the torch stand-in"""
    sd, m = _module(TINY)
    m._ops_backend = tob
    m.set_numframe(Fr)
    ctx = _randn((b, 77, TINY["channels"]), 3)
    got = m(context=ctx)
    ref = O.fstext_forward(sd, ctx, Fr, heads=TINY["n_heads"])
    assert got.shape == ref.shape == (b, Fr, 77, TINY["channels"])
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel


def test_no_cpu_fallback():
    from seervideoldm_amd._lib import SeerHipError
    _, m = _module(TINY)
    with pytest.raises(SeerHipError):
        m(context=_randn((1, 77, 192), 1))


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("Fr,b,l", [(6, 1, 77), (4, 2, 77), (5, 1, 20)])
def test_hip_matches_oracle_tiny(Fr, b, l):
    dev = torch.device("cuda:0")
    sd, m = _module(TINY, dev)
    m.set_numframe(Fr)
    ctx = _randn((b, l, TINY["channels"]), 7)
    got = m(context=ctx.to(dev)).cpu()
    ref = O.fstext_forward(sd, ctx, Fr, heads=TINY["n_heads"])
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel                       # bf16 storage of activations / weights vs the fp32 oracle
    if (Fr, b, l) == (6, 1, 77):                 # and against the reference's own output for the same input
        g = {k: torch.from_numpy(v) for k, v in np.load(G / "fstext_tiny.npz").items()}
        got2 = m(context=g["context"][:1].to(dev)).cpu()
        rel2 = ((got2 - g["y_F6"][:1]).norm() / g["y_F6"][:1].norm()).item()
        assert rel2 < 2e-2, rel2


@pytest.mark.gpu
def test_hip_full_size():
    """the real configuration: FSTextTransformer(num_frames=16, num_layers=8), set_numframe(12), CLIP [1, 77, 768]"""
    dev = torch.device("cuda:0")
    cfg = dict(num_frames=16, num_layers=8, channels=768, n_heads=8, cross_attention_dim=768)
    sd, m = _module(cfg, dev)
    m.set_numframe(12)
    ctx = _randn((1, 77, 768), 11)
    got = m(context=ctx.to(dev))
    assert got.shape == (1, 12, 77, 768) and got.dtype == torch.float32
    again = m(context=ctx.to(dev))
    assert torch.equal(got, again)               # deterministic kernels: bit-identical reruns
    ref = O.fstext_forward(sd, ctx, 12, heads=8)
    rel = ((got.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel
