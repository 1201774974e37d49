"""AutoencoderKL (decode side) -- host-side mirror of the diffusers==0.10.2 VAE surface the reference calls
(`vae.decode(z).sample`, utils/ddim_sampling_utils.py:39), over libseer_hip.so.

Arithmetic = the vendored twin `ldm/modules/diffusionmodules/model.py:462-568` (Decoder) preceded by `post_quant_conv`
(ldm/models/autoencoder.py:330-333): GroupNorm(32, eps 1e-6) + swish + 3x3 convs, one single-head attention (d = C) at
the lowest resolution, three nearest-2x upsamples each followed by a 3x3 conv.  Parameters use the diffusers key layout
(`decoder.mid_block.attentions.0.query`, ...) so an SD-v1-5 `vae/diffusion_pytorch_model.bin`
loads; `ldm_to_diffusers_vae` converts the vendored-ldm layout the CPU oracle uses.

Activations are channels-last bf16, accumulation fp32 (the reference runs its VAE in fp32: tolerance is stated in the
parity test).  The mid attention runs as batched MFMA GEMMs (q k^T -> fp32 scores -> row softmax -> p v) because d = 512
is outside the flash kernel's head dims; V is produced transposed by its projection GEMM's epilogue.
"""
from __future__ import annotations

from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops as hip_ops
from .unet import _build_tree, _Config
from .weights import pack_conv1x1, pack_conv3x3

bf16 = torch.bfloat16


def vae_decoder_shapes(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3):
    """diffusers key layout of post_quant_conv + decoder."""
    sh: "OrderedDict[str, tuple]" = OrderedDict()
    sh["post_quant_conv.weight"] = (latent_channels, latent_channels, 1, 1)
    sh["post_quant_conv.bias"] = (latent_channels,)

    def res(p, cin, cout):
        sh[p + ".norm1.weight"] = (cin,); sh[p + ".norm1.bias"] = (cin,)
        sh[p + ".conv1.weight"] = (cout, cin, 3, 3); sh[p + ".conv1.bias"] = (cout,)
        sh[p + ".norm2.weight"] = (cout,); sh[p + ".norm2.bias"] = (cout,)
        sh[p + ".conv2.weight"] = (cout, cout, 3, 3); sh[p + ".conv2.bias"] = (cout,)
        if cin != cout:
            sh[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); sh[p + ".conv_shortcut.bias"] = (cout,)

    D = "decoder."
    rev = list(reversed(block_out_channels))
    c = rev[0]
    sh[D + "conv_in.weight"] = (c, latent_channels, 3, 3); sh[D + "conv_in.bias"] = (c,)
    res(D + "mid_block.resnets.0", c, c)
    a = D + "mid_block.attentions.0"
    sh[a + ".group_norm.weight"] = (c,); sh[a + ".group_norm.bias"] = (c,)
    for nme in ("query", "key", "value", "proj_attn"):
        sh[f"{a}.{nme}.weight"] = (c, c); sh[f"{a}.{nme}.bias"] = (c,)
    res(D + "mid_block.resnets.1", c, c)
    for i, cout in enumerate(rev):
        for j in range(layers_per_block + 1):
            res(f"{D}up_blocks.{i}.resnets.{j}", c, cout)
            c = cout
        if i != len(rev) - 1:
            sh[f"{D}up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            sh[f"{D}up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    sh[D + "conv_norm_out.weight"] = (c,); sh[D + "conv_norm_out.bias"] = (c,)
    sh[D + "conv_out.weight"] = (out_channels, c, 3, 3); sh[D + "conv_out.bias"] = (out_channels,)
    return sh


def ldm_to_diffusers_vae(sd: Dict[str, torch.Tensor], n_levels: int) -> Dict[str, torch.Tensor]:
    """vendored-ldm decoder keys -> diffusers keys (same math, different names / [C,C,1,1] vs [C,C])."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.startswith("post_quant_conv."):
            out[k] = v
            continue
        assert k.startswith("decoder.")
        r = k[len("decoder."):]
        r = r.replace("mid.block_1.", "mid_block.resnets.0.").replace("mid.block_2.", "mid_block.resnets.1.")
        r = r.replace("nin_shortcut.", "conv_shortcut.").replace("norm_out.", "conv_norm_out.")
        if r.startswith("mid.attn_1."):
            leaf = r[len("mid.attn_1."):]
            name, wb = leaf.rsplit(".", 1)
            name = {"norm": "group_norm", "q": "query", "k": "key", "v": "value", "proj_out": "proj_attn"}[name]
            r = f"mid_block.attentions.0.{name}.{wb}"
            if name != "group_norm" and wb == "weight":
                v = v.reshape(v.shape[0], v.shape[1])
        elif r.startswith("up."):
            parts = r.split(".")
            lvl = int(parts[1])
            i = n_levels - 1 - lvl
            if parts[2] == "block":
                r = f"up_blocks.{i}.resnets.{parts[3]}." + ".".join(parts[4:])
            else:   # upsample.conv
                r = f"up_blocks.{i}.upsamplers.0." + ".".join(parts[3:])
        out["decoder." + r] = v
    return out


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, **ignored):
        super().__init__()
        self.config = _Config(in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              latent_channels=latent_channels, norm_num_groups=norm_num_groups)
        self._shapes = vae_decoder_shapes(block_out_channels, layers_per_block, latent_channels, out_channels)
        _build_tree(self, self._shapes)
        self._w: Optional[Dict[str, torch.Tensor]] = None

    def load_state_dict(self, state_dict, strict=True, **kw):
        # encoder / quant_conv keys of a full VAE checkpoint are not on the decode path: drop them
        sd = {k: v for k, v in state_dict.items() if k in self._shapes}
        self._w = None
        return super().load_state_dict(sd, strict=strict, **kw)

    def _apply(self, fn, *a, **k):
        self._w = None
        return super()._apply(fn, *a, **k)

    def encode(self, x):
        raise NotImplementedError("VAE encode is a 'next' row (SURVEY 8(f) rank 3); the hot path only decodes")

    # ---- packed weights ---------------------------------------------------------------------------------------------
    def prepare(self):
        dev = next(self.parameters()).device
        w = {}
        for k, v in self.state_dict().items():
            v = v.detach().to(dev, torch.float32)
            if k == "decoder.conv_in.weight":
                w[k] = v.permute(2, 3, 1, 0).contiguous()
            elif k == "decoder.conv_out.weight":
                w[k] = v.permute(0, 2, 3, 1).contiguous()
            elif k == "post_quant_conv.weight":
                w[k] = v.reshape(v.shape[0], v.shape[1]).contiguous()
            elif k.endswith(".weight") and v.dim() == 4:
                w[k] = (pack_conv3x3(v) if v.shape[-1] == 3 else pack_conv1x1(v)).to(bf16).contiguous()
            elif k.endswith(".weight") and v.dim() == 2:
                w[k] = v.to(bf16).contiguous()
            else:
                w[k] = v.contiguous()
        self._w = w
        self._dev = dev
        return self

    # ---- decode -----------------------------------------------------------------------------------------------------
    def _gn(self, x, N, rows, name, silu):
        ops = hip_ops
        G = self.config.norm_num_groups
        stats = torch.empty((N, G, 2), device=x.device, dtype=torch.float32)
        ops.groupnorm_stats(x, None, N, G, stats)
        return ops.groupnorm_apply(x, None, N, G, stats, rows * (x.shape[1] // G), 1e-6, self._w[name + ".weight"],
                                   self._w[name + ".bias"], silu)

    def _res(self, p, x, N, H, W):
        ops, w = hip_ops, self._w
        h = self._gn(x, N, H * W, p + ".norm1", True)
        h = ops.conv3x3(h, w[p + ".conv1.weight"], N, H, W, bias=w[p + ".conv1.bias"])
        h = self._gn(h, N, H * W, p + ".norm2", True)
        sc = x
        if (p + ".conv_shortcut.weight") in w:
            sc = ops.gemm(x, w[p + ".conv_shortcut.weight"], bias=w[p + ".conv_shortcut.bias"])
        return ops.conv3x3(h, w[p + ".conv2.weight"], N, H, W, bias=w[p + ".conv2.bias"], residual=sc)

    def _attn(self, p, x, N, H, W):
        """AttnBlock (model.py:178-202): softmax(q k^T / sqrt(C)) v, single head."""
        ops, w = hip_ops, self._w
        C, HW = x.shape[1], H * W
        h = self._gn(x, N, HW, p + ".group_norm", False)
        q = ops.gemm(h, w[p + ".query.weight"], bias=w[p + ".query.bias"]).reshape(N, HW, C)
        k = ops.gemm(h, w[p + ".key.weight"], bias=w[p + ".key.bias"]).reshape(N, HW, C)
        vt = ops.gemm_batched(h.reshape(N, HW, C), w[p + ".value.weight"], bias=w[p + ".value.bias"], trans_out=True)
        s = ops.gemm_batched(q, k, out_f32=True)                       # [N, HW, HW] fp32 scores
        pr = ops.softmax_rows(s, float(C) ** -0.5)                     # bf16 probabilities
        o = ops.gemm_batched(pr, vt).reshape(N * HW, C)                # p @ v
        return ops.gemm(o, w[p + ".proj_attn.weight"], bias=w[p + ".proj_attn.bias"], residual=x)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        if not z.is_cuda:
            raise hip_ops._lib.SeerHipError("AutoencoderKL.decode needs ROCm tensors: the HIP kernels are the only compute path")
        if self._w is None or self._dev != z.device:
            self.prepare()
        ops, w = hip_ops, self._w
        N, _, H, W = z.shape
        z = ops.conv1x1_nchw(z.float().contiguous(), w["post_quant_conv.weight"], w["post_quant_conv.bias"])
        D = "decoder."
        x = ops.conv_in(z.reshape(N, z.shape[1], 1, H, W), w[D + "conv_in.weight"], w[D + "conv_in.bias"])
        x = self._res(D + "mid_block.resnets.0", x, N, H, W)
        x = self._attn(D + "mid_block.attentions.0", x, N, H, W)
        x = self._res(D + "mid_block.resnets.1", x, N, H, W)
        nlev = len(self.config.block_out_channels)
        for i in range(nlev):
            for j in range(self.config.layers_per_block + 1):
                x = self._res(f"{D}up_blocks.{i}.resnets.{j}", x, N, H, W)
            if i != nlev - 1:
                p = f"{D}up_blocks.{i}.upsamplers.0.conv"
                x = ops.conv3x3(x, w[p + ".weight"], N, H, W, upsample=True, bias=w[p + ".bias"])
                H, W = 2 * H, 2 * W
        x = self._gn(x, N, H * W, D + "conv_norm_out", True)
        img = ops.conv_out(x, w[D + "conv_out.weight"], w[D + "conv_out.bias"], N, 1, H, W)
        img = img.reshape(N, img.shape[1], H, W)
        return SimpleNamespace(sample=img) if return_dict else (img,)
