"""FSTextTransformer -- host-side mirror of `seer.models.unet_3d_condition.FSTextTransformer` over libseer_hip.so.

The step BEFORE the denoising path: the CLIP text embedding `[b, 77, 768]` is decomposed into
per-frame sub-instruction embeddings `[b, F, 77, 768]`, the `context` every `SeerUNet` step cross-attends to
(inference_img.py:80-81,175).  Same constructor keywords, same `state_dict()` keys (`pytorch_model_1.bin` of a Seer
checkpoint loads with strict=True), same call `fstext(context=...) -> Tensor[b, F, l, C]`.

Per layer (unet_3d_condition.py:464-484; attention.py:153-180,328-427):
  block 0  tokens [b*F, l, C]: x += SelfAttn(LN1 x)  (no mask: `causal` only acts when `temporal`, attention.py:523-526)
           tokens [b, F*l, C]: x += CrossAttn(LN2 x, CLIP sequence);  x += GEGLU-FF(LN3 x)
  block 1  tokens [b*l, F, C]: x += causal SelfAttn over the frames with rotary(q, k; position = frame index), x += FF(LN3 x)
then a final LayerNorm.  Activations stay token-major bf16 [b*F*l, C] in (b, f, l) order for the whole stack; block 1
reads its sequences through strides (token stride = l rows) instead of permuting.  Kernels: the same bf16 MFMA GEMM
(fused q|k|v, GEGLU / rotary / bias / residual epilogues), LayerNorm and flash attention (head dim 96 = 768 / 8) as the
UNet.  No CPU path.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as hip_ops
from . import synth
from .unet import _build_tree
from .weights import interleave_geglu

bf16 = torch.bfloat16
MAX_LENGTH = 1024       # seer/models/unet_3d_condition.py:38


class FSTextTransformer(nn.Module):
    _supports_gradient_checkpointing = True

    def __init__(self, num_frames=None, in_channels=768, out_channels=768, n_heads=8, num_layers=2,
                 cross_attention_dim=768):
        super().__init__()
        if in_channels != out_channels:
            raise NotImplementedError("FSTextTransformer with in_channels != out_channels (vision_projection branch, "
                                      "attention.py:343-348) is not used by any Seer checkpoint")
        if out_channels % n_heads or (out_channels // n_heads) not in (40, 80, 96, 160):
            raise ValueError("head dim must be one of 40 / 80 / 96 / 160 (flash attention instantiations)")
        self.num_frames = num_frames
        self.channels, self.n_heads, self.num_layers, self.cross_attention_dim = out_channels, n_heads, num_layers, cross_attention_dim
        _build_tree(self, synth.fstext_param_shapes(num_frames, num_layers, out_channels, n_heads, cross_attention_dim,
                                                    MAX_LENGTH))
        self._w: Optional[Dict[str, torch.Tensor]] = None
        self._tokens = {}              # (b, F, l) -> bf16 [b*F*l, C] = learnable_query + pos_embed (input independent)
        self._rot = {}                 # (layer, F, l) -> rotary table expanded to one row per (f, l) token
        self._ops_backend = hip_ops    # tests may inject tests/torch_ops_backend.py (CPU host-logic tests)

    # ---- reference surface -------------------------------------------------------------------------------------------
    def set_numframe(self, num_frames):
        self.num_frames = num_frames

    def set_attention_slice(self, slice_size):
        return None                    # attention never materialises S x S here

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return self

    def _set_gradient_checkpointing(self, module, value=False):
        pass

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def _invalidate(self):
        self._w, self._tokens, self._rot = None, {}, {}

    # ---- packed weights ----------------------------------------------------------------------------------------------
    def prepare(self):
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        dev = next(self.parameters()).device
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        b16 = lambda t: t.to(dev, torch.float32).to(bf16).contiguous()
        w: Dict[str, torch.Tensor] = {}
        for n in range(self.num_layers):
            for d in (0, 1):
                p = f"trf_blocks.{n}.transformer_blocks.{d}"
                a1 = p + ".attn1"
                w[a1 + ".qkv"] = b16(torch.cat([sd[a1 + ".to_q.weight"], sd[a1 + ".to_k.weight"], sd[a1 + ".to_v.weight"]], 0))
                w[a1 + ".out.w"], w[a1 + ".out.b"] = b16(sd[a1 + ".to_out.0.weight"]), f32(sd[a1 + ".to_out.0.bias"])
                if d == 0:
                    a2 = p + ".attn2"
                    w[a2 + ".q"] = b16(sd[a2 + ".to_q.weight"])
                    w[a2 + ".kv"] = b16(torch.cat([sd[a2 + ".to_k.weight"], sd[a2 + ".to_v.weight"]], 0))
                    w[a2 + ".out.w"], w[a2 + ".out.b"] = b16(sd[a2 + ".to_out.0.weight"]), f32(sd[a2 + ".to_out.0.bias"])
                else:
                    w[a1 + ".freqs"] = f32(sd[a1 + ".rotary_emb.freqs"])
                wi, bi = interleave_geglu(sd[p + ".ff.net.0.proj.weight"], sd[p + ".ff.net.0.proj.bias"])
                w[p + ".ff1.w"], w[p + ".ff1.b"] = b16(wi), f32(bi)
                w[p + ".ff2.w"], w[p + ".ff2.b"] = b16(sd[p + ".ff.net.2.weight"]), f32(sd[p + ".ff.net.2.bias"])
                for nm in ("norm1", "norm3") + (("norm2",) if d == 0 else ()):
                    w[f"{p}.{nm}.w"], w[f"{p}.{nm}.b"] = f32(sd[f"{p}.{nm}.weight"]), f32(sd[f"{p}.{nm}.bias"])
        w["norm.w"], w["norm.b"] = f32(sd["norm.weight"]), f32(sd["norm.bias"])
        self._w = w
        return self

    def _token_init(self, b: int, Fr: int, l: int) -> torch.Tensor:
        """learnable_query + pos_embed[:, :, :l] (nearest resize over (frames, length) when the frame count differs,
        unet_3d_condition.py:468-474), replicated over the batch: input independent, cached like a packed weight."""
        key = (b, Fr, l)
        t = self._tokens.get(key)
        if t is None:
            pos = self.pos_embed.detach()[:, :, :l, :].float()
            if self.pos_embed.shape[1] != Fr:
                pos = F.interpolate(pos.permute(0, 3, 1, 2), size=(Fr, l)).permute(0, 2, 3, 1)
            x = (self.learnable_query.detach().float() + pos).expand(b, Fr, l, -1)
            t = x.reshape(b * Fr * l, -1).to(bf16).contiguous()
            self._tokens = {key: t}
        return t

    def _rot_rows(self, name: str, freqs: torch.Tensor, Fr: int, l: int) -> torch.Tensor:
        """cos/sin of position f for the row of token (f, l): the fused rotary epilogue indexes by row % (F*l)"""
        key = (name, Fr, l)
        t = self._rot.get(key)
        if t is None:
            t = self._ops_backend.rotary_table(freqs, Fr).repeat_interleave(l, dim=0).contiguous()
            self._rot[key] = t
        return t

    # ---- forward -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, context: torch.Tensor) -> torch.Tensor:
        ops = self._ops_backend
        if not context.is_cuda and ops is hip_ops:
            raise hip_ops._lib.SeerHipError("FSTextTransformer.forward needs ROCm tensors: the HIP kernels are the only compute path")
        if self._w is None:
            self.prepare()
        w = self._w
        b, l, cdim = context.shape
        Fr, C, heads = self.num_frames, self.channels, self.n_heads
        d = C // heads
        rot_dim = min(32, d)
        ctx = context.reshape(b * l, cdim)
        ctx = ops.cast_bf16(ctx.float().contiguous()) if ctx.dtype != bf16 else ctx.contiguous()
        x = self._token_init(b, Fr, l).clone()                 # [b*F*l, C], rows ordered (b, f, l)
        a = torch.empty_like(x)

        def ff(p):
            n3 = ops.layernorm(x, w[p + ".norm3.w"], w[p + ".norm3.b"])
            g = ops.gemm(n3, w[p + ".ff1.w"], bias=w[p + ".ff1.b"], geglu=True)
            ops.gemm(g, w[p + ".ff2.w"], bias=w[p + ".ff2.b"], residual=x, out=x)

        for n in range(self.num_layers):
            # ---- block 0: self-attention inside each frame's 77 tokens, cross-attention to the CLIP sequence, FF
            p = f"trf_blocks.{n}.transformer_blocks.0"
            n1 = ops.layernorm(x, w[p + ".norm1.w"], w[p + ".norm1.b"])
            qkv = ops.gemm(n1, w[p + ".attn1.qkv"])
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a, batch=b * Fr, heads=heads, head_dim=d, Sq=l, Sk=l)
            ops.gemm(a, w[p + ".attn1.out.w"], bias=w[p + ".attn1.out.b"], residual=x, out=x)
            n2 = ops.layernorm(x, w[p + ".norm2.w"], w[p + ".norm2.b"])
            q = ops.gemm(n2, w[p + ".attn2.q"])
            kv = ops.gemm(ctx, w[p + ".attn2.kv"])
            ops.attention(q, kv[:, :C], kv[:, C:], a, batch=b, heads=heads, head_dim=d, Sq=Fr * l, Sk=l)
            ops.gemm(a, w[p + ".attn2.out.w"], bias=w[p + ".attn2.out.b"], residual=x, out=x)
            ff(p)
            # ---- block 1: causal attention over the frames of each token position (rotary, position = frame), FF
            p = f"trf_blocks.{n}.transformer_blocks.1"
            n1 = ops.layernorm(x, w[p + ".norm1.w"], w[p + ".norm1.b"])
            cs = self._rot_rows(p, w[p + ".attn1.freqs"], Fr, l)
            qkv = ops.gemm(n1, w[p + ".attn1.qkv"], rotary=(cs, Fr * l, 0, d, rot_dim, 2 * C))
            for b0 in range(b):                               # sequence (b0, l0): rows b0*F*l + f*l + l0
                sl = slice(b0 * Fr * l, (b0 + 1) * Fr * l)
                ops.attention(qkv[sl, :C], qkv[sl, C:2 * C], qkv[sl, 2 * C:], a[sl], batch=l, heads=heads, head_dim=d,
                              Sq=Fr, Sk=Fr, causal=True, seq_stride_rows=l, batch_stride_rows=1)
            ops.gemm(a, w[p + ".attn1.out.w"], bias=w[p + ".attn1.out.b"], residual=x, out=x)
            ff(p)
        y = ops.layernorm(x, w["norm.w"], w["norm.b"])
        return y.float().reshape(b, Fr, l, C).to(context.dtype if context.dtype.is_floating_point else torch.float32)
