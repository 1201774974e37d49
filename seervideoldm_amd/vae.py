"""AutoencoderKL -- host-side mirror of the diffusers==0.10.2 VAE surface the reference calls, over libseer_hip.so:
`vae.decode(z).sample` on the path (utils/ddim_sampling_utils.py:39) and, before it, `vae.encode(x).latent_dist.sample()`
for the conditioning frames (inference_img.py:168).

Arithmetic = the vendored twins in `ldm/modules/diffusionmodules/model.py`: Decoder (:462-568) preceded by `post_quant_conv`,
Encoder (:368-460) followed by `quant_conv` (ldm/models/autoencoder.py:324-333) and DiagonalGaussianDistribution
(ldm/modules/distributions/distributions.py:24-37): GroupNorm(32, eps 1e-6) + swish + 3x3 convs, one single-head attention
(d = C) at the lowest resolution, nearest-2x upsamples followed by a 3x3 conv (decoder), stride-2 3x3 convs padded only
after the last row / column (encoder).  Parameters use the diffusers key layout (`decoder.mid_block.attentions.0.query`,
...) so an SD-v1-5 `vae/diffusion_pytorch_model.bin` loads; `ldm_to_diffusers_vae` converts the
vendored-ldm layout the CPU oracle uses.  A state dict may hold either half or both.

Precision.  The reference never autocasts its VAE (inference_img.py:118: the VAE is not `prepare`d; it decodes in fp32).
Activations and packed weights here are channels-last 16-bit with fp32 accumulation and fp32 GroupNorm / softmax
statistics; `compute_dtype` picks the storage type: torch.float16 (default: 11 significand bits -- 8x tighter than bf16 at
the same MFMA rate; the VAE's activations sit far inside fp16's range) or torch.bfloat16 (the UNet's type).  The parity
tests state the bound for each.  The mid attention runs as batched MFMA GEMMs (q k^T -> fp32 scores -> row softmax -> p v) because d = 512
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


def _res_shapes(sh, p, cin, cout):
    sh[p + ".norm1.weight"] = (cin,); sh[p + ".norm1.bias"] = (cin,)
    sh[p + ".conv1.weight"] = (cout, cin, 3, 3); sh[p + ".conv1.bias"] = (cout,)
    sh[p + ".norm2.weight"] = (cout,); sh[p + ".norm2.bias"] = (cout,)
    sh[p + ".conv2.weight"] = (cout, cout, 3, 3); sh[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        sh[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); sh[p + ".conv_shortcut.bias"] = (cout,)


def _mid_shapes(sh, P, c):
    _res_shapes(sh, P + "mid_block.resnets.0", c, c)
    a = P + "mid_block.attentions.0"
    sh[a + ".group_norm.weight"] = (c,); sh[a + ".group_norm.bias"] = (c,)
    for nme in ("query", "key", "value", "proj_attn"):
        sh[f"{a}.{nme}.weight"] = (c, c); sh[f"{a}.{nme}.bias"] = (c,)
    _res_shapes(sh, P + "mid_block.resnets.1", c, c)


def vae_decoder_shapes(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3):
    """diffusers key layout of post_quant_conv + decoder."""
    sh: "OrderedDict[str, tuple]" = OrderedDict()
    sh["post_quant_conv.weight"] = (latent_channels, latent_channels, 1, 1)
    sh["post_quant_conv.bias"] = (latent_channels,)
    D = "decoder."
    rev = list(reversed(block_out_channels))
    c = rev[0]
    sh[D + "conv_in.weight"] = (c, latent_channels, 3, 3); sh[D + "conv_in.bias"] = (c,)
    _mid_shapes(sh, D, c)
    for i, cout in enumerate(rev):
        for j in range(layers_per_block + 1):
            _res_shapes(sh, f"{D}up_blocks.{i}.resnets.{j}", c, cout)
            c = cout
        if i != len(rev) - 1:
            sh[f"{D}up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            sh[f"{D}up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    sh[D + "conv_norm_out.weight"] = (c,); sh[D + "conv_norm_out.bias"] = (c,)
    sh[D + "conv_out.weight"] = (out_channels, c, 3, 3); sh[D + "conv_out.bias"] = (out_channels,)
    return sh


def vae_encoder_shapes(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, in_channels=3):
    """diffusers key layout of encoder + quant_conv."""
    sh: "OrderedDict[str, tuple]" = OrderedDict()
    E = "encoder."
    c = block_out_channels[0]
    sh[E + "conv_in.weight"] = (c, in_channels, 3, 3); sh[E + "conv_in.bias"] = (c,)
    for i, cout in enumerate(block_out_channels):
        for j in range(layers_per_block):
            _res_shapes(sh, f"{E}down_blocks.{i}.resnets.{j}", c, cout)
            c = cout
        if i != len(block_out_channels) - 1:
            sh[f"{E}down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            sh[f"{E}down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    _mid_shapes(sh, E, c)
    sh[E + "conv_norm_out.weight"] = (c,); sh[E + "conv_norm_out.bias"] = (c,)
    sh[E + "conv_out.weight"] = (2 * latent_channels, c, 3, 3); sh[E + "conv_out.bias"] = (2 * latent_channels,)
    sh["quant_conv.weight"] = (2 * latent_channels, 2 * latent_channels, 1, 1)
    sh["quant_conv.bias"] = (2 * latent_channels,)
    return sh


def ldm_to_diffusers_vae(sd: Dict[str, torch.Tensor], n_levels: int) -> Dict[str, torch.Tensor]:
    """vendored-ldm encoder / decoder keys -> diffusers keys (same math, different names,
    [C,C,1,1] vs [C,C] attention projections, decoder levels counted from the other end)."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.startswith(("post_quant_conv.", "quant_conv.")):
            out[k] = v
            continue
        side = "decoder." if k.startswith("decoder.") else "encoder."
        assert k.startswith(side), k
        r = k[len(side):]
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
            i = n_levels - 1 - int(parts[1])
            if parts[2] == "block":
                r = f"up_blocks.{i}.resnets.{parts[3]}." + ".".join(parts[4:])
            else:   # upsample.conv
                r = f"up_blocks.{i}.upsamplers.0." + ".".join(parts[3:])
        elif r.startswith("down."):
            parts = r.split(".")
            if parts[2] == "block":
                r = f"down_blocks.{parts[1]}.resnets.{parts[3]}." + ".".join(parts[4:])
            else:   # downsample.conv
                r = f"down_blocks.{parts[1]}.downsamplers.0." + ".".join(parts[3:])
        out[side + r] = v
    return out


class DiagonalGaussianDistribution:
    """the `latent_dist` of `AutoencoderKL.encode` (diffusers 0.10.2 surface; arithmetic of
    ldm/modules/distributions/distributions.py:24-37)."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments                       # fp32 [N, 2C, h, w] = (mean | logvar)
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=torch.float32)
        return hip_ops.gaussian_sample(self.parameters, noise)

    def mode(self) -> torch.Tensor:
        return hip_ops.gaussian_sample(self.parameters, None)


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, compute_dtype=torch.float16, **ignored):
        super().__init__()
        if compute_dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("AutoencoderKL.compute_dtype: torch.float16 or torch.bfloat16")
        self.compute_dtype = compute_dtype
        self.config = _Config(in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              latent_channels=latent_channels, norm_num_groups=norm_num_groups)
        self._dec_shapes = vae_decoder_shapes(block_out_channels, layers_per_block, latent_channels, out_channels)
        self._enc_shapes = vae_encoder_shapes(block_out_channels, layers_per_block, latent_channels, in_channels)
        self._shapes = OrderedDict(list(self._enc_shapes.items()) + list(self._dec_shapes.items()))
        _build_tree(self, self._shapes)
        self._w: Optional[Dict[str, torch.Tensor]] = None
        self._loaded = {"encoder": False, "decoder": False}

    config_name = "config.json"

    @classmethod
    def from_pretrained(cls, path, subfolder=None, revision=None, **kw):
        """diffusers semantics (inference_img.py:69: `AutoencoderKL.from_pretrained(model, subfolder="vae")`): read
        config.json (unknown keys ignored), load `diffusion_pytorch_model.{safetensors,bin}` by name."""
        import inspect
        import json
        import os
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = json.load(f)
        allowed = set(inspect.signature(cls.__init__).parameters) - {"self", "ignored"}
        model = cls(**{k: v for k, v in cfg.items() if k in allowed})
        for fname in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin"):
            fp = os.path.join(root, fname)
            if os.path.exists(fp):
                if fname.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(fp)
                else:
                    sd = torch.load(fp, map_location="cpu")
                model.load_state_dict(sd, strict=True)
                return model
        raise FileNotFoundError(f"no weight file under {root}")

    def load_state_dict(self, state_dict, strict=True, **kw):
        """a checkpoint may carry the decoder half, the encoder half or both (the hot path only needs the decoder);
        `strict` applies to each half that is present."""
        sd = {k: v for k, v in state_dict.items() if k in self._shapes}
        has = {"encoder": any(k in self._enc_shapes for k in sd), "decoder": any(k in self._dec_shapes for k in sd)}
        if strict:
            for half, shapes in (("encoder", self._enc_shapes), ("decoder", self._dec_shapes)):
                missing = [k for k in shapes if k not in sd]
                if has[half] and missing:
                    raise RuntimeError(f"AutoencoderKL.load_state_dict: {half} keys missing: {missing[:4]}...")
            if not (has["encoder"] or has["decoder"]):
                raise RuntimeError("AutoencoderKL.load_state_dict: no VAE keys in the state dict")
        self._w = None
        out = super().load_state_dict(sd, strict=False, **kw)
        for half in has:
            self._loaded[half] = self._loaded[half] or has[half]
        return out

    def _apply(self, fn, *a, **k):
        self._w = None
        return super()._apply(fn, *a, **k)

    # ---- packed weights ---------------------------------------------------------------------------------------------
    def prepare(self):
        dev = next(self.parameters()).device
        w = {}
        bf16 = self.compute_dtype                      # storage type of every packed 16-bit weight below
        for k, v in self.state_dict().items():
            v = v.detach().to(dev, torch.float32)
            if k in ("decoder.conv_in.weight", "encoder.conv_in.weight"):
                w[k] = v.permute(2, 3, 1, 0).contiguous()                   # [3,3,Cin,Cout] fp32: direct conv_in kernel
            elif k == "decoder.conv_out.weight":
                w[k] = v.permute(0, 2, 3, 1).contiguous()                   # Cout = 3: direct conv_out kernel
            elif k == "encoder.conv_out.weight":
                w[k] = pack_conv3x3(v).to(bf16).contiguous()                # Cout = 8: MFMA conv_out route
            elif k in ("post_quant_conv.weight", "quant_conv.weight"):
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

    def _ready(self, t: torch.Tensor, half: str):
        if not t.is_cuda:
            raise hip_ops._lib.SeerHipError("AutoencoderKL needs ROCm tensors: the HIP kernels are the only compute path")
        if not self._loaded[half]:
            raise RuntimeError(f"AutoencoderKL: no {half} weights were loaded")
        if self._w is None or self._dev != t.device:
            self.prepare()

    # ---- shared blocks ----------------------------------------------------------------------------------------------
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
        if HW % 64 == 0:
            pr = ops.softmax_rows(s, float(C) ** -0.5, dtype=self.compute_dtype)     # 16-bit probabilities
        else:       # small / odd latents (e.g. 8x12): the p @ v contraction needs a multiple of 64 -> zero-padded keys
            HWp = (HW + 63) // 64 * 64
            pr = torch.zeros((N, HW, HWp), device=x.device, dtype=self.compute_dtype)
            ops.softmax_rows(s, float(C) ** -0.5, out=pr)
            vp = torch.zeros((N, C, HWp), device=x.device, dtype=self.compute_dtype)
            vp[:, :, :HW] = vt
            vt = vp
        o = ops.gemm_batched(pr, vt).reshape(N * HW, C)                # p @ v
        return ops.gemm(o, w[p + ".proj_attn.weight"], bias=w[p + ".proj_attn.bias"], residual=x)

    def _mid(self, P, x, N, H, W):
        x = self._res(P + "mid_block.resnets.0", x, N, H, W)
        x = self._attn(P + "mid_block.attentions.0", x, N, H, W)
        return self._res(P + "mid_block.resnets.1", x, N, H, W)

    # ---- encode (before the path) -------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [N, 3, H, W] in [-1, 1] -> `.latent_dist` (DiagonalGaussianDistribution over [N, 4, H/8, W/8]); the caller
        scales the sample by 0.18215 (inference_img.py:168-169)."""
        self._ready(x, "encoder")
        ops, w = hip_ops, self._w
        N, Cin, H, W = x.shape
        nlev = len(self.config.block_out_channels)
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)):
            raise ValueError("image height/width must be multiples of 2^(levels-1)")
        E = "encoder."
        h = ops.conv_in(x.float().reshape(N, Cin, 1, H, W).contiguous(), w[E + "conv_in.weight"], w[E + "conv_in.bias"],
                        dtype=self.compute_dtype)
        for i in range(nlev):
            for j in range(self.config.layers_per_block):
                h = self._res(f"{E}down_blocks.{i}.resnets.{j}", h, N, H, W)
            if i != nlev - 1:       # F.pad(x, (0,1,0,1)) + conv(stride 2, padding 0) (model.py:60-78)
                p = f"{E}down_blocks.{i}.downsamplers.0.conv"
                h = ops.conv3x3(h, w[p + ".weight"], N, H, W, stride=2, pad_after_only=True, bias=w[p + ".bias"])
                H, W = H // 2, W // 2
        h = self._mid(E, h, N, H, W)
        h = self._gn(h, N, H * W, E + "conv_norm_out", True)
        m = ops.conv_out(h, w[E + "conv_out.weight"], w[E + "conv_out.bias"], N, 1, H, W)      # [N, 2z, 1, h, w] fp32
        m = ops.conv1x1_nchw(m.reshape(N, m.shape[1], H, W), w["quant_conv.weight"], w["quant_conv.bias"])
        dist = DiagonalGaussianDistribution(m)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    # ---- decode (on the path) -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        self._ready(z, "decoder")
        ops, w = hip_ops, self._w
        N, _, H, W = z.shape
        z = ops.conv1x1_nchw(z.float().contiguous(), w["post_quant_conv.weight"], w["post_quant_conv.bias"])
        D = "decoder."
        x = ops.conv_in(z.reshape(N, z.shape[1], 1, H, W), w[D + "conv_in.weight"], w[D + "conv_in.bias"], dtype=self.compute_dtype)
        x = self._mid(D, x, N, H, W)
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
