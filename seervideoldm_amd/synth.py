"""Architecture parameter inventory + closed-form synthetic weights.

`unet_param_shapes(cfg)` enumerates the reference's `SeerUNet.state_dict()` keys/shapes (SURVEY 8(b) "Weights") from the
config alone; `vae_param_shapes()` does the same for the SD VAE decoder in the vendored-ldm key layout.
`synth_state_dict` fills them with an RNG-free closed-form function of (key name, element index), so the reference
(in the build container), the CPU oracle and the HIP path can all be given bit-identical weights without shipping a
checkpoint (there is no network for real ones).  `proj_out` is NOT zero (the reference zero-inits it,
attention.py:127, which would silence every transformer in a parity test).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

Shapes = "OrderedDict[str, Tuple[int, ...]]"

SD15_UNET_CFG = dict(sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True,
                     freq_shift=0, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                     downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
                     cross_attention_dim=768, attention_head_dim=8)


def _resnet(sh, p, cin, cout, temb):
    sh[p + ".norm1.weight"] = (cin,); sh[p + ".norm1.bias"] = (cin,)
    sh[p + ".conv1.weight"] = (cout, cin, 3, 3); sh[p + ".conv1.bias"] = (cout,)
    sh[p + ".time_emb_proj.weight"] = (cout, temb); sh[p + ".time_emb_proj.bias"] = (cout,)
    sh[p + ".norm2.weight"] = (cout,); sh[p + ".norm2.bias"] = (cout,)
    sh[p + ".conv2.weight"] = (cout, cout, 3, 3); sh[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        sh[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); sh[p + ".conv_shortcut.bias"] = (cout,)


def _attn(sh, p, C, ctx_dim, heads, rotary):
    if rotary:
        sh[p + ".rotary_emb.freqs"] = (min(32, C // heads) // 2,)
    sh[p + ".to_q.weight"] = (C, C)
    sh[p + ".to_k.weight"] = (C, ctx_dim)
    sh[p + ".to_v.weight"] = (C, ctx_dim)
    sh[p + ".to_out.0.weight"] = (C, C); sh[p + ".to_out.0.bias"] = (C,)


def _transformer(sh, p, C, ctx_dim, heads, temporal):
    sh[p + ".norm.weight"] = (C,); sh[p + ".norm.bias"] = (C,)
    sh[p + ".proj_in.weight"] = (C, C, 1, 1); sh[p + ".proj_in.bias"] = (C,)
    tb = p + ".transformer_blocks.0"
    _attn(sh, tb + ".attn1", C, C, heads, rotary=temporal)
    sh[tb + ".ff.net.0.proj.weight"] = (8 * C, C); sh[tb + ".ff.net.0.proj.bias"] = (8 * C,)
    sh[tb + ".ff.net.2.weight"] = (C, 4 * C); sh[tb + ".ff.net.2.bias"] = (C,)
    if not temporal:
        _attn(sh, tb + ".attn2", C, ctx_dim, heads, rotary=False)
        sh[tb + ".norm2.weight"] = (C,); sh[tb + ".norm2.bias"] = (C,)
    sh[tb + ".norm1.weight"] = (C,); sh[tb + ".norm1.bias"] = (C,)
    sh[tb + ".norm3.weight"] = (C,); sh[tb + ".norm3.bias"] = (C,)
    sh[p + ".proj_out.weight"] = (C, C, 1, 1); sh[p + ".proj_out.bias"] = (C,)


def unet_param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """keys/shapes of SeerUNet.state_dict() (seer/models/unet_3d_condition.py:64-205, unet_3d_blocks.py)."""
    c = dict(SD15_UNET_CFG); c.update(cfg)
    boc = tuple(c["block_out_channels"]); lpb = c["layers_per_block"]; heads = c["attention_head_dim"]
    ctx = c["cross_attention_dim"]; temb = boc[0] * 4
    n = len(boc)
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    sh["conv_in.weight"] = (boc[0], c["in_channels"], 3, 3); sh["conv_in.bias"] = (boc[0],)
    sh["time_embedding.linear_1.weight"] = (temb, boc[0]); sh["time_embedding.linear_1.bias"] = (temb,)
    sh["time_embedding.linear_2.weight"] = (temb, temb); sh["time_embedding.linear_2.bias"] = (temb,)
    out = boc[0]
    for i in range(n):
        inp, out = out, boc[i]
        p = f"down_blocks.{i}"
        for j in range(lpb):
            _resnet(sh, f"{p}.resnets.{j}", inp if j == 0 else out, out, temb)
            if i < n - 1:
                _transformer(sh, f"{p}.attentions.{j}", out, ctx, heads, False)
                _transformer(sh, f"{p}.temporal_attentions.{j}", out, ctx, heads, True)
        if i < n - 1:
            sh[f"{p}.downsamplers.0.conv.weight"] = (out, out, 3, 3); sh[f"{p}.downsamplers.0.conv.bias"] = (out,)
    C = boc[-1]
    _resnet(sh, "mid_block.resnets.0", C, C, temb)
    _transformer(sh, "mid_block.attentions.0", C, ctx, heads, False)
    _transformer(sh, "mid_block.temporal_attentions.0", C, ctx, heads, True)
    _resnet(sh, "mid_block.resnets.1", C, C, temb)
    rev = list(reversed(boc))
    out = rev[0]
    for i in range(n):
        prev, out = out, rev[i]
        inp = rev[min(i + 1, n - 1)]
        p = f"up_blocks.{i}"
        for j in range(lpb + 1):
            skip = inp if j == lpb else out
            rin = prev if j == 0 else out
            _resnet(sh, f"{p}.resnets.{j}", rin + skip, out, temb)
            if i > 0:
                _transformer(sh, f"{p}.attentions.{j}", out, ctx, heads, False)
                _transformer(sh, f"{p}.temporal_attentions.{j}", out, ctx, heads, True)
        if i < n - 1:
            sh[f"{p}.upsamplers.0.conv.weight"] = (out, out, 3, 3); sh[f"{p}.upsamplers.0.conv.bias"] = (out,)
    sh["conv_norm_out.weight"] = (boc[0],); sh["conv_norm_out.bias"] = (boc[0],)
    sh["conv_out.weight"] = (c["out_channels"], boc[0], 3, 3); sh["conv_out.bias"] = (c["out_channels"],)
    return sh


def fstext_param_shapes(num_frames=16, num_layers=8, channels=768, n_heads=8, cross_attention_dim=768, max_length=1024):
    """FSTextTransformer state dict (seer/models/unet_3d_condition.py:379-400; blocks attention.py:153-180,328-362):
    `pytorch_model_1.bin` of a Seer checkpoint (inference_img.py:80,100-101: num_frames=16, num_layers=8)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    C = channels
    sh["learnable_query"] = (1, 1, 1, C)
    sh["pos_embed"] = (1, num_frames, max_length, C)
    for n in range(num_layers):
        for d, temporal in ((0, False), (1, True)):
            tb = f"trf_blocks.{n}.transformer_blocks.{d}"
            _attn(sh, tb + ".attn1", C, C, n_heads, rotary=temporal)
            sh[tb + ".ff.net.0.proj.weight"] = (8 * C, C); sh[tb + ".ff.net.0.proj.bias"] = (8 * C,)
            sh[tb + ".ff.net.2.weight"] = (C, 4 * C); sh[tb + ".ff.net.2.bias"] = (C,)
            if not temporal:
                _attn(sh, tb + ".attn2", C, cross_attention_dim, n_heads, rotary=False)
                sh[tb + ".norm2.weight"] = (C,); sh[tb + ".norm2.bias"] = (C,)
            sh[tb + ".norm1.weight"] = (C,); sh[tb + ".norm1.bias"] = (C,)
            sh[tb + ".norm3.weight"] = (C,); sh[tb + ".norm3.bias"] = (C,)
    sh["norm.weight"] = (C,); sh["norm.bias"] = (C,)
    return sh


def vae_param_shapes(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, out_ch=3):
    """SD VAE decoder in the vendored-ldm key layout (ldm/modules/diffusionmodules/model.py:462-533) + post_quant_conv."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    sh["post_quant_conv.weight"] = (z_channels, z_channels, 1, 1); sh["post_quant_conv.bias"] = (z_channels,)
    P = "decoder."

    def res(p, cin, cout):
        sh[p + ".norm1.weight"] = (cin,); sh[p + ".norm1.bias"] = (cin,)
        sh[p + ".conv1.weight"] = (cout, cin, 3, 3); sh[p + ".conv1.bias"] = (cout,)
        sh[p + ".norm2.weight"] = (cout,); sh[p + ".norm2.bias"] = (cout,)
        sh[p + ".conv2.weight"] = (cout, cout, 3, 3); sh[p + ".conv2.bias"] = (cout,)
        if cin != cout:
            sh[p + ".nin_shortcut.weight"] = (cout, cin, 1, 1); sh[p + ".nin_shortcut.bias"] = (cout,)

    block_in = ch * ch_mult[-1]
    sh[P + "conv_in.weight"] = (block_in, z_channels, 3, 3); sh[P + "conv_in.bias"] = (block_in,)
    res(P + "mid.block_1", block_in, block_in)
    a = P + "mid.attn_1"
    sh[a + ".norm.weight"] = (block_in,); sh[a + ".norm.bias"] = (block_in,)
    for nme in ("q", "k", "v", "proj_out"):
        sh[f"{a}.{nme}.weight"] = (block_in, block_in, 1, 1); sh[f"{a}.{nme}.bias"] = (block_in,)
    res(P + "mid.block_2", block_in, block_in)
    for lvl in reversed(range(len(ch_mult))):
        block_out = ch * ch_mult[lvl]
        for j in range(num_res_blocks + 1):
            res(f"{P}up.{lvl}.block.{j}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            sh[f"{P}up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            sh[f"{P}up.{lvl}.upsample.conv.bias"] = (block_in,)
    sh[P + "norm_out.weight"] = (block_in,); sh[P + "norm_out.bias"] = (block_in,)
    sh[P + "conv_out.weight"] = (out_ch, block_in, 3, 3); sh[P + "conv_out.bias"] = (out_ch,)
    return sh


def vae_encoder_param_shapes(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, in_channels=3):
    """SD VAE encoder in the vendored-ldm key layout (ldm/modules/diffusionmodules/model.py:368-431) + quant_conv."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    P = "encoder."

    def res(p, cin, cout):
        sh[p + ".norm1.weight"] = (cin,); sh[p + ".norm1.bias"] = (cin,)
        sh[p + ".conv1.weight"] = (cout, cin, 3, 3); sh[p + ".conv1.bias"] = (cout,)
        sh[p + ".norm2.weight"] = (cout,); sh[p + ".norm2.bias"] = (cout,)
        sh[p + ".conv2.weight"] = (cout, cout, 3, 3); sh[p + ".conv2.bias"] = (cout,)
        if cin != cout:
            sh[p + ".nin_shortcut.weight"] = (cout, cin, 1, 1); sh[p + ".nin_shortcut.bias"] = (cout,)

    sh[P + "conv_in.weight"] = (ch, in_channels, 3, 3); sh[P + "conv_in.bias"] = (ch,)
    block_in = ch
    for lvl, m in enumerate(ch_mult):
        block_out = ch * m
        for j in range(num_res_blocks):
            res(f"{P}down.{lvl}.block.{j}", block_in, block_out)
            block_in = block_out
        if lvl != len(ch_mult) - 1:
            sh[f"{P}down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            sh[f"{P}down.{lvl}.downsample.conv.bias"] = (block_in,)
    res(P + "mid.block_1", block_in, block_in)
    a = P + "mid.attn_1"
    sh[a + ".norm.weight"] = (block_in,); sh[a + ".norm.bias"] = (block_in,)
    for nme in ("q", "k", "v", "proj_out"):
        sh[f"{a}.{nme}.weight"] = (block_in, block_in, 1, 1); sh[f"{a}.{nme}.bias"] = (block_in,)
    res(P + "mid.block_2", block_in, block_in)
    sh[P + "norm_out.weight"] = (block_in,); sh[P + "norm_out.bias"] = (block_in,)
    sh[P + "conv_out.weight"] = (2 * z_channels, block_in, 3, 3); sh[P + "conv_out.bias"] = (2 * z_channels,)
    sh["quant_conv.weight"] = (2 * z_channels, 2 * z_channels, 1, 1); sh["quant_conv.bias"] = (2 * z_channels,)
    return sh


def _hash_uniform(name: str, n: int, device) -> torch.Tensor:
    """closed-form u[i] in [0,1): frac(sin(i*a + h(name)) * b), float64 arithmetic."""
    h = (zlib.crc32(name.encode()) % 100003) * 0.001
    i = torch.arange(n, dtype=torch.float64, device=device)
    v = torch.sin(i * 12.9898 + h) * 43758.5453
    return v - torch.floor(v)


def synth_tensor(name: str, shape, device="cpu", gain: float = 1.0) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= s
    if name.endswith("rotary_emb.freqs"):
        dim = 2 * shape[0]
        return (1.0 / (10000 ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))).to(device)
    u = _hash_uniform(name, n, device)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = any(t in name for t in (".norm", "norm1", "norm2", "norm3", "norm_out", "group_norm")) or name.startswith("norm.")
    if leaf in ("learnable_query", "pos_embed"):      # FSTextTransformer tokens: O(1) values (zero-initialised in the reference)
        val = u - 0.5
    elif is_norm and leaf == "weight":
        val = 1.0 + 0.2 * (u - 0.5)
    elif leaf == "bias":
        val = 0.1 * (u - 0.5)
    else:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        val = (2.0 * u - 1.0) * math.sqrt(3.0) * gain / math.sqrt(max(fan_in, 1))
    return val.to(torch.float32).reshape(shape)


def synth_state_dict(shapes, device="cpu", gain: float = 1.0) -> Dict[str, torch.Tensor]:
    return OrderedDict((k, synth_tensor(k, s, device, gain)) for k, s in shapes.items())
