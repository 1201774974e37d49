"""HIP path against fixtures produced by the REAL reference at kernel-supported widths (oracle/make_goldens_w320.py): one
hop, no oracle in between.  Weights are the closed-form synthetic ones (a function of the parameter name), inputs and
expected outputs come from tests/golden/*_w320*.npz."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from seervideoldm_amd import SeerUNet, synth

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
bf16 = torch.bfloat16
CALIB = json.loads((GOLD / "calibration_bf16.json").read_text())["unet_w320_bf16_autocast_vs_fp32"]
# what the reference itself loses when it runs under bf16 autocast (measured by the generating script on the same network and
# inputs): 1.8e-2.  The HIP path keeps bf16 activations end to end; its bound is 1.65 x that number (= tests/test_gpu_unet.py)
REL_L2 = 1.65 * CALIB["rel_l2"]

W320_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(320, 320, 320, 320),
                 cross_attention_dim=256, attention_head_dim=8, layers_per_block=1)


def _load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(GOLD / name).items()}


def _rel(got, ref):
    got, ref = got.float().cpu(), ref.float()
    assert torch.isfinite(got).all()
    return ((got - ref).norm() / ref.norm()).item()


def test_unet_w320_against_the_reference(device):
    g = _load("unet_w320_real.npz")
    m = SeerUNet(**W320_UNET)
    m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(W320_UNET)), strict=True)
    m = m.to(device).eval()
    x, t, ctx = g["sample"].to(device), g["timestep"].to(device), g["context"].to(device)
    for cond, key in ((0, "y_cond0"), (1, "y_cond1")):
        rel = _rel(m(x, t, ctx, cond_frame=cond), g[key])
        print(f"[parity vs reference] SeerUNet w320 cond_frame={cond}: rel_l2 {rel:.4g} (bound {REL_L2:.3g}, "
              f"reference bf16 autocast {CALIB['rel_l2']:.3g})")
        assert rel <= REL_L2


def _w(name, shape, device):
    return synth.synth_tensor(name, shape).to(device)


def test_self_attention_w320_against_the_reference(device):
    """CrossAttention(query_dim=320, heads=8, dim_head=40) as self-attention over 1024 tokens (attention.py:429-630):
    [8, 1024, 40] per batch element through the fused q|k|v projection, the d = 40 attention kernel and to_out"""
    from seervideoldm_amd import ops
    g = _load("op_selfattn_w320.npz")
    C = 320
    x = g["x"].reshape(-1, C).to(device).to(bf16)
    wqkv = torch.cat([_w(f"w320.attn1.to_{n}.weight", (C, C), device) for n in "qkv"]).to(bf16)
    qkv = ops.gemm(x, wqkv, col_scale=(ops.qk_prescale(40), C))
    a = torch.empty_like(x)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a, batch=g["x"].shape[0], heads=8, head_dim=40,
                  Sq=1024, Sk=1024, q_prescaled=True)
    y = ops.gemm(a, _w("w320.attn1.to_out.0.weight", (C, C), device).to(bf16),
                 bias=_w("w320.attn1.to_out.0.bias", (C,), device))
    rel = _rel(y, g["y"].reshape(-1, C))
    print(f"[parity vs reference] self-attention w320: rel_l2 {rel:.4g}")
    assert rel <= 1e-2


def test_cross_attention_w320_against_the_reference(device):
    from seervideoldm_amd import ops
    g = _load("op_crossattn_w320.npz")
    C, Dc = 320, 768
    B = g["x"].shape[0]
    x = g["x"].reshape(-1, C).to(device).to(bf16)
    ctx = g["context"].reshape(-1, Dc).to(device).to(bf16)
    q = ops.gemm(x, _w("w320.attn2.to_q.weight", (C, C), device).to(bf16))
    wkv = torch.cat([_w(f"w320.attn2.to_{n}.weight", (C, Dc), device) for n in "kv"]).to(bf16)
    kv = ops.gemm(ctx, wkv)
    a = torch.empty_like(x)
    ops.attention(q, kv[:, :C], kv[:, C:], a, batch=B, heads=8, head_dim=40, Sq=1024, Sk=77)
    y = ops.gemm(a, _w("w320.attn2.to_out.0.weight", (C, C), device).to(bf16),
                 bias=_w("w320.attn2.to_out.0.bias", (C,), device))
    rel = _rel(y, g["y"].reshape(-1, C))
    print(f"[parity vs reference] text cross-attention w320: rel_l2 {rel:.4g}")
    assert rel <= 1e-2


def test_feedforward_w320_against_the_reference(device):
    """FeedForward(320) = GEGLU projection (value | gate, exact-erf GELU) + Linear (attention.py:705-793)"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import geglu_row_order
    g = _load("op_feedforward_w320.npz")
    C, Hd = 320, 1280
    x = g["x"].to(device).to(bf16)
    order = geglu_row_order(Hd).to(device)
    w1 = _w("w320.ff.net.0.proj.weight", (2 * Hd, C), device)[order].to(bf16)
    b1 = _w("w320.ff.net.0.proj.bias", (2 * Hd,), device)[order].contiguous()
    h = ops.gemm(x, w1.contiguous(), bias=b1, geglu=True)
    y = ops.gemm(h, _w("w320.ff.net.2.weight", (C, Hd), device).to(bf16), bias=_w("w320.ff.net.2.bias", (C,), device))
    rel = _rel(y, g["y"])
    print(f"[parity vs reference] feed-forward w320: rel_l2 {rel:.4g}")
    assert rel <= 1e-2
