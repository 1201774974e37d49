"""SeerTrainer -- the reference's fine-tuning step (train.py:319-389) over libseer_hip.so.

What trains (train.py:122-124,188-192,213): every parameter under a `*.temporal_attentions` module of the SeerUNet and the
whole FSTextTransformer; the rest of the UNet, the VAE and CLIP are frozen.  One step =
    text_seq = fstext(text_cond_emb)                                       train.py:344
    pred     = sunet(cat[latents_x0, noisy_latents], t, text_seq, cond)    train.py:367
    loss     = mse(pred[:, :, cond:], noise)                               train.py:369-380
    backward; clip_grad_norm_(sunet params, max_grad_norm); AdamW          train.py:382-387

Design (MI355X): no autograd tape -- the forward keeps exactly the activations its hand-written backward needs (bf16,
token-major) and the backward walks the schedule in reverse, issuing the same GEMM / conv / attention kernels on
transposed operands plus the HBM-bound kernels of csrc/train.hip.  Frozen layers only propagate dX; trainable layers also
produce dW (fp32, straight into one flat gradient buffer).  Parameters live in flat fp32 master / Adam-moment buffers in
the PACKED layouts the kernels read (fused q|k|v, interleaved GEGLU rows): AdamW is element-wise, so one launch per
segment updates everything and refreshes the bf16 working copy; `trainable_state_dict()` unpacks to the reference's names.
Data parallel: one all-reduce of the flat gradient buffer (RCCL) before the optimizer.

No CPU path: the only arithmetic backends are libseer_hip.so's entry points (tests inject a torch stand-in to check this
file's wiring on CPU against autograd of the oracle).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops as hip_ops
from . import train_ops as hip_train_ops
from .fstext import FSTextTransformer
from .unet import MAX_RATIO, MAX_WIN_SIZE, MIN_WIN_SIZE, SeerUNet, _Engine
from .weights import geglu_row_order

bf16 = torch.bfloat16
f32 = torch.float32


def cosine_lr(step: int, base_lr: float, warmup_steps: int, total_steps: int) -> float:
    """diffusers.optimization.get_cosine_schedule_with_warmup (`lr_scheduler: "cosine"`, train.py:258-263): linear warm-up to
    base_lr over warmup_steps, then half a cosine to zero at total_steps.  `step` = optimizer steps taken so far."""
    import math
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * 2.0 * 0.5 * progress)))


class _Params:
    """flat fp32 master / gradient / Adam buffers with named views (packed layouts)."""

    def __init__(self, tensors: "Dict[str, torch.Tensor]", device):
        self.names = list(tensors)
        self.shapes = {k: tuple(v.shape) for k, v in tensors.items()}
        self.offsets, off = {}, 0
        for k, v in tensors.items():
            self.offsets[k] = off
            off += (v.numel() + 7) // 8 * 8            # keep every view 16/32-byte aligned
        self.n = off
        self.p = torch.zeros((off,), device=device, dtype=f32)
        for k, v in tensors.items():
            self.view(self.p, k).copy_(v.to(device, f32))
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.pb = self.p.to(bf16)

    def view(self, flat: torch.Tensor, k: str) -> torch.Tensor:
        o = self.offsets[k]
        n = 1
        for s in self.shapes[k]:
            n *= s
        return flat[o:o + n].view(self.shapes[k])


def _pack_temporal_fp32(sd: Dict[str, torch.Tensor]) -> "Dict[str, torch.Tensor]":
    """trainable UNet tensors (fp32) in the engine's packed layouts and under the engine's names (unet._Engine._pack)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if ".temporal_attentions." not in k or k.endswith("rotary_emb.freqs"):
            continue
        v = v.detach().float()
        if k.endswith((".to_k.weight", ".to_v.weight")):
            continue
        if k.endswith(".to_q.weight"):
            p = k[: -len(".to_q.weight")]
            out[p + ".qkv"] = torch.cat([v, sd[p + ".to_k.weight"].detach().float(), sd[p + ".to_v.weight"].detach().float()], 0)
        elif k.endswith("ff.net.0.proj.weight"):
            out[k] = v[geglu_row_order(v.shape[0] // 2)]
        elif k.endswith("ff.net.0.proj.bias"):
            out[k] = v[geglu_row_order(v.shape[0] // 2)]
        elif v.dim() == 4:
            out[k] = v.reshape(v.shape[0], v.shape[1])
        else:
            out[k] = v
    return out


def _pack_fstext_fp32(sd: Dict[str, torch.Tensor], num_layers: int) -> "Dict[str, torch.Tensor]":
    """FSTextTransformer tensors (fp32) under the names of FSTextTransformer.prepare() + the two embeddings."""
    g = lambda k: sd[k].detach().float()
    out: Dict[str, torch.Tensor] = {"learnable_query": g("learnable_query"), "pos_embed": g("pos_embed")}
    for n in range(num_layers):
        for d in (0, 1):
            p = f"trf_blocks.{n}.transformer_blocks.{d}"
            a1 = p + ".attn1"
            out[a1 + ".qkv"] = torch.cat([g(a1 + ".to_q.weight"), g(a1 + ".to_k.weight"), g(a1 + ".to_v.weight")], 0)
            out[a1 + ".out.w"], out[a1 + ".out.b"] = g(a1 + ".to_out.0.weight"), g(a1 + ".to_out.0.bias")
            if d == 0:
                a2 = p + ".attn2"
                out[a2 + ".q"] = g(a2 + ".to_q.weight")
                out[a2 + ".kv"] = torch.cat([g(a2 + ".to_k.weight"), g(a2 + ".to_v.weight")], 0)
                out[a2 + ".out.w"], out[a2 + ".out.b"] = g(a2 + ".to_out.0.weight"), g(a2 + ".to_out.0.bias")
            w1, b1 = g(p + ".ff.net.0.proj.weight"), g(p + ".ff.net.0.proj.bias")
            order = geglu_row_order(w1.shape[0] // 2)
            out[p + ".ff1.w"], out[p + ".ff1.b"] = w1[order], b1[order]
            out[p + ".ff2.w"], out[p + ".ff2.b"] = g(p + ".ff.net.2.weight"), g(p + ".ff.net.2.bias")
            for nm in ("norm1", "norm3") + (("norm2",) if d == 0 else ()):
                out[f"{p}.{nm}.w"], out[f"{p}.{nm}.b"] = g(f"{p}.{nm}.weight"), g(f"{p}.{nm}.bias")
    out["norm.w"], out["norm.b"] = g("norm.weight"), g("norm.bias")
    return out


class SeerTrainer:
    def __init__(self, unet: SeerUNet, fstext: FSTextTransformer, *, lr: float = 1e-4, betas=(0.9, 0.999),
                 weight_decay: float = 1e-2, eps: float = 1e-8, max_grad_norm: float = 1.0, gradient_accumulation_steps: int = 1,
                 text_loss: bool = False, ops=hip_ops, tops=hip_train_ops, process_group=None):
        self.ops, self.tops = ops, tops
        self.accum = int(gradient_accumulation_steps)           # configs/train.yaml:13, train.py:321 (accelerator.accumulate)
        self._micro = 0
        self.text_loss = bool(text_loss)                       # train.py:346-347,377-378 (configs/train.yaml text_loss)
        self.lr, self.betas, self.weight_decay, self.eps, self.max_grad_norm = lr, betas, weight_decay, eps, max_grad_norm
        self.pg = process_group
        self.unet, self.fstext = unet, fstext
        self.eng = _Engine(unet, ops=ops, fold_ln=False)
        self.device = self.eng.device
        self.step_count = 0
        sd_u = {k: v for k, v in unet.state_dict().items()}
        sd_f = {k: v for k, v in fstext.state_dict().items()}
        self.pu = _Params(_pack_temporal_fp32(sd_u), self.device)
        self.pf = _Params(_pack_fstext_fp32(sd_f, fstext.num_layers), self.device)
        for P in (self.pu, self.pf):                            # mean of the micro-batch gradients when accumulating
            P.acc = torch.zeros_like(P.g) if self.accum > 1 else None
        # working weights: bf16 views for matrices, fp32 master views for biases / norm affine; the frozen rest comes from
        # the engine's packed dict
        self.w: Dict[str, torch.Tensor] = dict(self.eng.w)
        self.trainable_u = set(self.pu.names)
        for k in self.pu.names:
            self.w[k] = self.pu.view(self.pu.pb if len(self.pu.shapes[k]) == 2 else self.pu.p, k)
        self.wf: Dict[str, torch.Tensor] = {}
        for k in self.pf.names:
            is_mat = len(self.pf.shapes[k]) == 2
            self.wf[k] = self.pf.view(self.pf.pb if is_mat else self.pf.p, k)
        for n in range(fstext.num_layers):
            fk = f"trf_blocks.{n}.transformer_blocks.1.attn1.rotary_emb.freqs"
            self.wf[f"trf_blocks.{n}.transformer_blocks.1.attn1.freqs"] = sd_f[fk].detach().to(self.device, f32).contiguous()
        self.conv_out_w = sd_u["conv_out.weight"].detach().to(self.device, f32).permute(0, 2, 3, 1).contiguous()
        # W^T of every trainable matrix (the dX products read the weight [K][N]): one arena, refreshed by ONE launch at the top of
        # each step (the weights move with every AdamW step) instead of one transpose launch per layer inside the backward pass
        mats = [(P, W, k) for P, W in ((self.pu, self.w), (self.pf, self.wf)) for k in P.names if len(P.shapes[k]) == 2]
        self._wt_plan = tops.TransposePlan([W[k] for _, W, k in mats])
        for P in (self.pu, self.pf):
            P.wT = {}
        for (P, _, k), y in zip(mats, self._wt_plan.outputs):
            P.wT[k] = y
        self._wT: Dict[str, torch.Tensor] = {}          # transposed copies of frozen matrices (dX GEMMs), built on first use
        self._rot_conj: Dict[int, torch.Tensor] = {}
        self._kv_cols = None
        self._fs_rot: Dict[Tuple, torch.Tensor] = {}    # FSTextTransformer rotary tables, one row per (f, l) token
        self._pos_src: Dict[int, torch.Tensor] = {}     # F -> source frame of pos_embed per output frame (device)
        self._graphs: Dict[Tuple, Tuple] = {}           # captured forward+backward, keyed on the input shapes
        self._graph_broken = False
        self._early_hu = None                           # handle of a UNet-segment all-reduce started inside the step
        # weight gradients: deferred to the end of each backward walk and launched as a group (SEER_DW_GROUPED=0: one launch per layer
        # where the walk reaches it)
        self._dw_deferred = os.environ.get("SEER_DW_GROUPED", "1") != "0" and hasattr(tops, "gemm_tn_grouped")
        self._dw: List[Tuple] = []
        self._cf: Optional[List[Tuple]] = [] if self._dw_deferred and hasattr(tops, "colfinal_grouped") else None   # d gamma / d beta slabs of the LayerNorms, as the weight gradients

    # ================================================================================================ helpers
    def _frozenT(self, key: str) -> torch.Tensor:
        t = self._wT.get(key)
        if t is None:
            t = self.tops.transpose(self.w[key])
            self._wT[key] = t
        return t

    def _convT(self, key: str) -> torch.Tensor:
        """[Co, 9*Ci] packed conv weight -> the input-gradient conv's weight [Ci, 9*Co]: w'[ci][ky][kx][co] = w[co][2-ky][2-kx][ci]"""
        t = self._wT.get(key)
        if t is None:
            w = self.w[key]
            Co = w.shape[0]
            Ci = w.shape[1] // 9
            t = w.reshape(Co, 3, 3, Ci).flip(1, 2).permute(3, 1, 2, 0).reshape(Ci, 9 * Co).contiguous()
            self._wT[key] = t
        return t

    def _conj(self, cs: torch.Tensor) -> torch.Tensor:
        t = self._rot_conj.get(cs.data_ptr())
        if t is None:
            t = cs.clone()
            t[..., 1].neg_()
            self._rot_conj[cs.data_ptr()] = t
        return t

    def _lin_bwd(self, P: Optional[_Params], W: Dict[str, torch.Tensor], wkey: str, bkey: Optional[str], x: Optional[torch.Tensor],
                 dy: torch.Tensor, *, need_dx=True, dres=None, out=None):
        """y = x W^T + b.  P != None: the layer trains (dW, db into P.g).  Returns dx (+ dres) or None."""
        ops, tops = self.ops, self.tops
        if P is not None:
            prob = (dy, x, P.view(P.g, wkey), P.view(P.g, bkey) if bkey is not None else None)
            if self._dw_deferred:       # nothing reads a weight gradient before the optimizer: all of a pass's products in one launch
                self._dw.append(prob)   # (_flush_dw); dy and x are kept alive, and nothing writes to them, until then
            else:
                tops.gemm_tn(prob[0], prob[1], out=prob[2], colsum=prob[3])
        if not need_dx:
            return None
        WT = P.wT[wkey] if P is not None else self._frozenT(wkey)
        return ops.gemm(dy, WT, residual=dres, out=out)

    def _flush_dw(self):
        """the weight-gradient products the backward walk left behind (`_lin_bwd`), as one grouped launch"""
        if self._dw:
            self.tops.gemm_tn_grouped(self._dw)
            self._dw = []
        if self._cf:
            self.tops.colfinal_grouped(self._cf)
            self._cf = []

    def _cb(self, B):
        """`colsum_batch` of a launch whose output feeds a GroupNorm: accumulate into the step's arena when there is one"""
        return (B, self._fx) if getattr(self, "_fx", None) is not None else B

    def _gn_fwd(self, x1, x2, B, rows_pb, name, eps, silu):
        ops, w = self.ops, self.w
        stats = torch.empty((B, self.eng.G, 2), device=x1.device, dtype=f32)
        # as in the inference engine (unet._Engine._gn): the column sums the producing GEMM / conv left next to its output, when
        # every source has them -- no statistics pass over the activations
        cs1 = getattr(x1, "colsums", None)
        cs2 = getattr(x2, "colsums", None) if x2 is not None else None
        C = x1.shape[1] + (0 if x2 is None else x2.shape[1])
        count = rows_pb * (C // self.eng.G)
        FX = getattr(ops, "ColSumsFx", ())
        if isinstance(cs1, FX) and (x2 is None or isinstance(cs2, FX)):
            # accumulated fixed-point sums (unet._Engine._gn): one launch normalises and leaves the statistics for the backward
            y = ops.groupnorm_apply_fx(x1, x2, cs1, cs2, B, self.eng.G, count, eps, w[name + ".weight"], w[name + ".bias"], silu,
                                       stats_out=stats)
            if y is not None:
                return y, (x1, x2, B, stats, count, name, eps, silu)
            ops.groupnorm_stats_from_fx(cs1, cs2, B, self.eng.G, stats)
        elif isinstance(cs1, FX) or isinstance(cs2, FX):
            ops.groupnorm_stats(x1, x2, B, self.eng.G, stats)
        elif cs1 is not None and (x2 is None or cs2 is not None):
            ops.groupnorm_stats_from_colsums(cs1, cs2, B, self.eng.G, stats)
        else:
            ops.groupnorm_stats(x1, x2, B, self.eng.G, stats)
        y = ops.groupnorm_apply(x1, x2, B, self.eng.G, stats, count, eps, w[name + ".weight"], w[name + ".bias"], silu)
        return y, (x1, x2, B, stats, count, name, eps, silu)

    def _gn_bwd(self, saved, dy, dres1=None, dres2=None):
        x1, x2, B, stats, count, name, eps, silu = saved
        train = (name + ".weight") in self.trainable_u
        dg = self.pu.view(self.pu.g, name + ".weight") if train else None
        db = self.pu.view(self.pu.g, name + ".bias") if train else None
        return self.tops.groupnorm_bwd(x1, x2, B, self.eng.G, stats, count, eps, self.w[name + ".weight"], self.w[name + ".bias"],
                                       silu, dy, dres1=dres1, dres2=dres2, dgamma=dg, dbeta=db)

    # ================================================================================================ UNet blocks
    def _resnet_fwd(self, p, x, skip, geo):
        ops, w = self.ops, self.w
        B, Fr, H, W = geo
        rows_pb = Fr * H * W
        off, n = self.eng.temb_slices[p]
        temb = self._temb[:, off:off + n]
        h1, s1 = self._gn_fwd(x, skip, B, rows_pb, p + ".norm1", self.eng.eps, True)
        h2 = ops.conv3x3(h1, w[p + ".conv1.weight"], B * Fr, H, W, bias=w[p + ".conv1.bias"], rowvec=temb, rows_per_batch=rows_pb,
                         colsum_batch=self._cb(B))
        h3, s2 = self._gn_fwd(h2, None, B, rows_pb, p + ".norm2", self.eng.eps, True)
        has_sc = (p + ".conv_shortcut.weight") in w
        sc = ops.gemm(x, w[p + ".conv_shortcut.weight"], a2=skip, bias=w[p + ".conv_shortcut.bias"]) if has_sc else x
        out = ops.conv3x3(h3, w[p + ".conv2.weight"], B * Fr, H, W, bias=w[p + ".conv2.bias"], residual=sc, colsum_batch=self._cb(B))
        return out, (p, geo, s1, s2, has_sc, x.shape[1])

    def _resnet_bwd(self, saved, dout):
        ops = self.ops
        p, geo, s1, s2, has_sc, C1 = saved
        B, Fr, H, W = geo
        dh3 = ops.conv3x3(dout, self._convT(p + ".conv2.weight"), B * Fr, H, W)
        dh2, _ = self._gn_bwd(s2, dh3)
        dh1 = ops.conv3x3(dh2, self._convT(p + ".conv1.weight"), B * Fr, H, W)
        if has_sc:
            WT = self._frozenT(p + ".conv_shortcut.weight")          # [Cin, Cout]
            d1 = ops.gemm(dout, WT[:C1])
            d2 = ops.gemm(dout, WT[C1:]) if WT.shape[0] > C1 else None
        else:
            d1, d2 = dout, None
        return self._gn_bwd(s1, dh1, dres1=d1, dres2=d2)             # (dx, dskip)

    def _ff_fwd(self, P, W, names, hf, out=None):
        """GEGLU feed-forward on rows hf: returns (hf + FF(LN3 hf), saved); `out`: where to write the result"""
        ops, tops = self.ops, self.tops
        g3, b3, w1, b1, w2, b2 = names
        n3 = ops.layernorm(hf, W[g3], W[b3])
        pre = ops.gemm(n3, W[w1], bias=W[b1])
        g = tops.geglu_fwd(pre)
        out = ops.gemm(g, W[w2], bias=W[b2], residual=hf, out=out)
        return out, (hf, n3, pre, g if P is not None else None)     # a trainable ff.net.2 reads g again (its weight gradient)

    def _ff_bwd(self, P, W, names, saved, dout, dx=None):
        """returns d hf (LayerNorm path + residual path); `dx`: where to write it"""
        tops = self.tops
        g3, b3, w1, b1, w2, b2 = names
        hf, n3, pre, g = saved
        dg = self._lin_bwd(P, W, w2, b2, g, dout)
        dpre = tops.geglu_bwd(pre, dg)
        dn3 = self._lin_bwd(P, W, w1, b1, n3, dpre)
        return tops.layernorm_bwd(hf, dn3, W[g3], dres=dout, dx=dx,
                                  dgamma=P.view(P.g, g3) if P is not None else None,
                                  dbeta=P.view(P.g, b3) if P is not None else None, defer=self._cf)

    @staticmethod
    def _unet_ff_names(tb):
        return (tb + ".norm3.weight", tb + ".norm3.bias", tb + ".ff.net.0.proj.weight", tb + ".ff.net.0.proj.bias",
                tb + ".ff.net.2.weight", tb + ".ff.net.2.bias")

    def _text_fwd(self, p, x, geo):
        ops, tops, w = self.ops, self.tops, self.w
        B, Fr, H, W = geo
        C = x.shape[1]
        heads, d = self.eng.heads, C // self.eng.heads
        HW = H * W
        tb = p + ".transformer_blocks.0"
        hn, sg = self._gn_fwd(x, None, B, Fr * HW, p + ".norm", 1e-6, False)
        h0 = ops.gemm(hn, w[p + ".proj_in.weight"], bias=w[p + ".proj_in.bias"])
        n1 = ops.layernorm(h0, w[tb + ".norm1.weight"], w[tb + ".norm1.bias"])
        qkv = ops.gemm(n1, w[tb + ".attn1.qkv"])
        a1 = torch.empty_like(h0)
        kw1 = dict(batch=B * Fr, heads=heads, head_dim=d, Sq=HW, Sk=HW)
        lse1 = tops.attn_lse_buffer(B * Fr, heads, HW, x.device)
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a1, lse=lse1, **kw1)
        h1 = ops.gemm(a1, w[tb + ".attn1.to_out.0.weight"], bias=w[tb + ".attn1.to_out.0.bias"], residual=h0)
        n2 = ops.layernorm(h1, w[tb + ".norm2.weight"], w[tb + ".norm2.bias"])
        q2 = ops.gemm(n2, w[tb + ".attn2.q"])
        kv = ops.gemm(self._ctx, w[tb + ".attn2.kv"])
        a2 = torch.empty_like(h0)
        L = self._ctx_len
        kw2 = dict(batch=B * Fr, heads=heads, head_dim=d, Sq=HW, Sk=L)
        lse2 = tops.attn_lse_buffer(B * Fr, heads, HW, x.device)
        ops.attention(q2, kv[:, :C], kv[:, C:], a2, lse=lse2, **kw2)
        h2 = ops.gemm(a2, w[tb + ".attn2.to_out.0.weight"], bias=w[tb + ".attn2.to_out.0.bias"], residual=h1)
        if (p + ".ffproj.w") in w:
            # the text blocks are frozen: ff.net.2 and proj_out as ONE two-source GEMM over [h2 | g] (the engine's folded pair,
            # unet._Engine._pack: x + [Wp | Wp W2] [h2 | g] + (Wp b2 + bp)) -- the backward needs neither h3 nor their weights apart
            g3, b3, w1, b1, _, _ = self._unet_ff_names(tb)
            n3 = ops.layernorm(h2, w[g3], w[b3])
            pre = ops.gemm(n3, w[w1], bias=w[b1])
            g = tops.geglu_fwd(pre)
            out = ops.gemm(h2, w[p + ".ffproj.w"], a2=g, bias=w[p + ".ffproj.b"], residual=x, colsum_batch=self._cb(B))
            sff = (h2, n3, pre, None)
        else:
            h3, sff = self._ff_fwd(None, w, self._unet_ff_names(tb), h2)
            out = ops.gemm(h3, w[p + ".proj_out.weight"], bias=w[p + ".proj_out.bias"], residual=x, colsum_batch=self._cb(B))
        return out, (p, C, sg, h0, qkv, a1, lse1, kw1, h1, q2, kv, a2, lse2, kw2, sff)

    def _text_bwd(self, saved, dout, stop_after_kv=False):
        ops, tops, w = self.ops, self.tops, self.w
        p, C, sg, h0, qkv, a1, lse1, kw1, h1, q2, kv, a2, lse2, kw2, sff = saved
        tb = p + ".transformer_blocks.0"
        if (p + ".ffproj.w") in w:
            # [d h3 | d g] = dout [Wp | Wp W2]: the two input gradients of the folded pair from ONE GEMM
            g3, _, w1, _, _, _ = self._unet_ff_names(tb)
            hf, n3, pre, _ = sff
            both = ops.gemm(dout, self._frozenT(p + ".ffproj.w"))
            dpre = tops.geglu_bwd(pre, both[:, C:])
            dn3 = self._lin_bwd(None, w, w1, None, None, dpre)
            dh2 = tops.layernorm_bwd(hf, dn3, w[g3], dres=both[:, :C])
        else:
            dh3 = self._lin_bwd(None, w, p + ".proj_out.weight", None, None, dout)
            dh2 = self._ff_bwd(None, w, self._unet_ff_names(tb), sff, dh3)
        # text cross attention: dK|dV go straight into this block's column range of the shared [B*F*L, sum 2C] buffer
        da2 = self._lin_bwd(None, w, tb + ".attn2.to_out.0.weight", None, None, dh2)
        c0 = self._kv_cols[tb]
        dkv = self._dkv_all[:, c0:c0 + 2 * C]
        dq2 = torch.empty_like(q2)
        tops.attention_bwd(q2, kv[:, :C], kv[:, C:], a2, lse2, da2, dq2, dkv[:, :C], dkv[:, C:], **kw2)
        if stop_after_kv:
            return None
        dn2 = self._lin_bwd(None, w, tb + ".attn2.q", None, None, dq2)
        dh1 = tops.layernorm_bwd(h1, dn2, w[tb + ".norm2.weight"], dres=dh2)
        da1 = self._lin_bwd(None, w, tb + ".attn1.to_out.0.weight", None, None, dh1)
        dqkv = torch.empty_like(qkv)
        tops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a1, lse1, da1, dqkv[:, :C], dqkv[:, C:2 * C],
                           dqkv[:, 2 * C:], **kw1)
        dn1 = self._lin_bwd(None, w, tb + ".attn1.qkv", None, None, dqkv)
        dh0 = tops.layernorm_bwd(h0, dn1, w[tb + ".norm1.weight"], dres=dh1)
        dhn = self._lin_bwd(None, w, p + ".proj_in.weight", None, None, dh0)
        dx, _ = self._gn_bwd(sg, dhn, dres1=dout)
        return dx

    @staticmethod
    def _ff_slices(B, Fr, HW, skip_f):
        return [slice(b * Fr * HW + skip_f * HW, (b + 1) * Fr * HW) for b in range(B)]

    @staticmethod
    def _gather(t, sl):
        return t[sl[0]] if len(sl) == 1 else torch.cat([t[s] for s in sl], 0)

    @staticmethod
    def _scatter(dst, src, sl):
        o = 0
        for s in sl:
            n = s.stop - s.start
            dst[s].copy_(src[o:o + n])
            o += n

    def _temporal_fwd(self, p, x, geo, cond_frame):
        ops, tops, w = self.ops, self.tops, self.w
        B, Fr, H, W = geo
        C = x.shape[1]
        heads, d = self.eng.heads, C // self.eng.heads
        HW = H * W
        tb = p + ".transformer_blocks.0"
        hn, sg = self._gn_fwd(x, None, B, Fr * HW, p + ".norm", 1e-6, False)
        h0 = ops.gemm(hn, w[p + ".proj_in.weight"], bias=w[p + ".proj_in.bias"])
        n1 = ops.layernorm(h0, w[tb + ".norm1.weight"], w[tb + ".norm1.bias"])
        rot_dim = min(32, d)
        cs = self.eng._rotary_table(tb, Fr * HW)
        qkv = ops.gemm(n1, w[tb + ".attn1.qkv"], rotary=(cs, Fr * HW, 0, d, rot_dim, 2 * C))
        a1 = torch.empty_like(h0)
        if H > MIN_WIN_SIZE:
            ws = MAX_WIN_SIZE if (H // MAX_WIN_SIZE) >= MAX_RATIO else MIN_WIN_SIZE
            kw = dict(batch=B, heads=heads, head_dim=d, Sq=Fr * ws * ws, Sk=Fr * ws * ws, causal=True, window=(ws, Fr, H, W))
            lse = tops.attn_lse_buffer(B, heads, Fr * ws * ws, x.device, window=(ws, Fr, H, W))
        else:
            kw = dict(batch=B, heads=heads, head_dim=d, Sq=Fr * HW, Sk=Fr * HW, causal=True)
            lse = tops.attn_lse_buffer(B, heads, Fr * HW, x.device)
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a1, lse=lse, **kw)
        h1 = ops.gemm(a1, w[tb + ".attn1.to_out.0.weight"], bias=w[tb + ".attn1.to_out.0.bias"], residual=h0)
        sl = self._ff_slices(B, Fr, HW, cond_frame) if cond_frame > 0 else [slice(0, B * Fr * HW)]
        if cond_frame > 0 and B == 1:        # FF rows are one contiguous slice: write them in place, copy only the cond rows
            h2 = torch.empty_like(h1)
            h2[:sl[0].start].copy_(h1[:sl[0].start])
            _, sff = self._ff_fwd(self.pu, w, self._unet_ff_names(tb), h1[sl[0]], out=h2[sl[0]])
        else:
            hf2, sff = self._ff_fwd(self.pu, w, self._unet_ff_names(tb), self._gather(h1, sl))
            if cond_frame > 0:
                h2 = h1.clone()
                self._scatter(h2, hf2, sl)
            else:
                h2 = hf2
        out = ops.gemm(h2, w[p + ".proj_out.weight"], bias=w[p + ".proj_out.bias"], residual=x, colsum_batch=self._cb(B))
        return out, (p, C, d, rot_dim, cs, Fr * HW, sg, hn, h0, n1, qkv, a1, lse, kw, sl, cond_frame, sff, h2)

    def _temporal_bwd(self, saved, dout):
        ops, tops, w, P = self.ops, self.tops, self.w, self.pu
        p, C, d, rot_dim, cs, tpb, sg, hn, h0, n1, qkv, a1, lse, kw, sl, cond_frame, sff, h2 = saved
        tb = p + ".transformer_blocks.0"
        dh2 = self._lin_bwd(P, w, p + ".proj_out.weight", p + ".proj_out.bias", h2, dout)
        if cond_frame > 0 and len(sl) == 1:
            dh1 = torch.empty_like(dh2)
            dh1[:sl[0].start].copy_(dh2[:sl[0].start])
            self._ff_bwd(P, w, self._unet_ff_names(tb), sff, dh2[sl[0]], dx=dh1[sl[0]])
        else:
            d_hf = self._ff_bwd(P, w, self._unet_ff_names(tb), sff, self._gather(dh2, sl))
            if cond_frame > 0:
                dh1 = dh2.clone()
                self._scatter(dh1, d_hf, sl)
            else:
                dh1 = d_hf
        da1 = self._lin_bwd(P, w, tb + ".attn1.to_out.0.weight", tb + ".attn1.to_out.0.bias", a1, dh1)
        dqkv = torch.empty_like(qkv)
        tops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a1, lse, da1, dqkv[:, :C], dqkv[:, C:2 * C],
                           dqkv[:, 2 * C:], **kw)
        ops.rotary_inplace(dqkv, 0, C, self.eng.heads, d, rot_dim, tpb, self._conj(cs))      # R^T on dq, dk
        dn1 = self._lin_bwd(P, w, tb + ".attn1.qkv", None, n1, dqkv)
        dh0 = tops.layernorm_bwd(h0, dn1, w[tb + ".norm1.weight"], dres=dh1, dgamma=P.view(P.g, tb + ".norm1.weight"),
                                 dbeta=P.view(P.g, tb + ".norm1.bias"), defer=self._cf)
        dhn = self._lin_bwd(P, w, p + ".proj_in.weight", p + ".proj_in.bias", hn, dh0)
        dx, _ = self._gn_bwd(sg, dhn, dres1=dout)
        return dx

    # ================================================================================================ UNet schedule
    def _text_blocks(self) -> List[str]:
        boc, lpb, n = self.eng.boc, self.eng.lpb, len(self.eng.boc)
        names = []
        for i in range(n - 1):
            names += [f"down_blocks.{i}.attentions.{j}" for j in range(lpb)]
        names.append("mid_block.attentions.0")
        for i in range(1, n):
            names += [f"up_blocks.{i}.attentions.{j}" for j in range(lpb + 1)]
        return names

    def _unet_fwd(self, sample, t, ctx_bf16, ctx_len, cond_frame):
        """the schedule of unet._Engine._forward, keeping what the backward needs.  Returns (pred fp32 [B,4,F,H,W], tape)."""
        ops, w, eng = self.ops, self.w, self.eng
        B, Cin, Fr, H, W = sample.shape
        boc, lpb, n = eng.boc, eng.lpb, len(eng.boc)
        self._ctx, self._ctx_len = ctx_bf16, ctx_len
        if self._kv_cols is None:
            self._kv_cols, c = {}, 0
            for name in self._text_blocks():
                lvl_c = w[name + ".proj_in.weight"].shape[0]
                self._kv_cols[name + ".transformer_blocks.0"] = c
                c += 2 * lvl_c
            self._kv_total = c
        self._dkv_all = torch.empty((ctx_bf16.shape[0], self._kv_total), device=sample.device, dtype=bf16)
        emb = ops.timestep_embedding(t, boc[0], eng.cfg.flip_sin_to_cos, eng.cfg.freq_shift)
        emb = ops.linear_smallm(emb, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"], silu_out=True)
        emb = ops.linear_smallm(emb, w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])
        self._temb = ops.linear_smallm(emb, w["temb_all.w"], w["temb_all.b"], silu_in=True)
        tape: List[Tuple] = []
        self._fx = None
        if getattr(self, "gn_fx", os.environ.get("SEER_GN_FX", "1") != "0") and hasattr(ops, "FxArena"):
            need = (eng.n_groupnorms() + 16) * B * 4 * max(boc) * 2
            if getattr(self, "_fx_arena", None) is None or self._fx_arena.buf.numel() < need:
                assert not (sample.is_cuda and torch.cuda.is_current_stream_capturing()), \
                    "the accumulator arena must exist before a graph capture"
                if getattr(self, "_fx_arena", None) is not None:
                    self._fx_retired = getattr(self, "_fx_retired", []) + [self._fx_arena]     # captured steps keep theirs by address
                self._fx_arena = ops.FxArena(sample.device, need)
            self._fx_arena.reset()
            self._fx = self._fx_arena
        x = ops.conv_in(sample, w["conv_in.weight"], w["conv_in.bias"])
        skips = [x]
        geo = (B, Fr, H, W)
        for i in range(n):
            p = f"down_blocks.{i}"
            for j in range(lpb):
                x, s = self._resnet_fwd(f"{p}.resnets.{j}", x, None, geo); tape.append(("resnet", s))
                if i < n - 1:
                    x, s = self._text_fwd(f"{p}.attentions.{j}", x, geo); tape.append(("text", s))
                    x, s = self._temporal_fwd(f"{p}.temporal_attentions.{j}", x, geo, cond_frame); tape.append(("temporal", s))
                skips.append(x); tape.append(("push", None))
            if i < n - 1:
                x = ops.conv3x3(x, w[f"{p}.downsamplers.0.conv.weight"], B * Fr, geo[2], geo[3], stride=2,
                                bias=w[f"{p}.downsamplers.0.conv.bias"], colsum_batch=self._cb(B))
                tape.append(("down", (f"{p}.downsamplers.0.conv.weight", geo)))
                geo = (B, Fr, (geo[2] - 1) // 2 + 1, (geo[3] - 1) // 2 + 1)
                skips.append(x); tape.append(("push", None))
        x, s = self._resnet_fwd("mid_block.resnets.0", x, None, geo); tape.append(("resnet", s))
        x, s = self._text_fwd("mid_block.attentions.0", x, geo); tape.append(("text", s))
        x, s = self._temporal_fwd("mid_block.temporal_attentions.0", x, geo, cond_frame); tape.append(("temporal", s))
        x, s = self._resnet_fwd("mid_block.resnets.1", x, None, geo); tape.append(("resnet", s))
        for i in range(n):
            p = f"up_blocks.{i}"
            for j in range(lpb + 1):
                x, s = self._resnet_fwd(f"{p}.resnets.{j}", x, skips.pop(), geo); tape.append(("resnet_pop", s))
                if i > 0:
                    x, s = self._text_fwd(f"{p}.attentions.{j}", x, geo); tape.append(("text", s))
                    x, s = self._temporal_fwd(f"{p}.temporal_attentions.{j}", x, geo, cond_frame); tape.append(("temporal", s))
            if i < n - 1:
                x = ops.conv3x3(x, w[f"{p}.upsamplers.0.conv.weight"], B * Fr, geo[2], geo[3], upsample=True,
                                bias=w[f"{p}.upsamplers.0.conv.bias"], colsum_batch=self._cb(B))
                tape.append(("up", (f"{p}.upsamplers.0.conv.weight", geo)))
                geo = (B, Fr, geo[2] * 2, geo[3] * 2)
        x, s = self._gn_fwd(x, None, B, Fr * geo[2] * geo[3], "conv_norm_out", eng.eps, True); tape.append(("gn_out", s))
        pred = ops.conv_out(x, w["conv_out.weight"], w["conv_out.bias"], B, Fr, geo[2], geo[3])
        return pred, tape

    def _unet_bwd(self, tape, dpred):
        """walks the tape in reverse; fills pu.g and the shared dK|dV buffer; returns d context (bf16 [B*F*L, Dc])"""
        ops, tops = self.ops, self.tops
        dx = tops.conv_out_bwd(dpred, self.conv_out_w)
        dskips: List[torch.Tensor] = []                 # gradients of the skip tensors, in the order they will be needed
        first_trainable = next(i for i, (k, _) in enumerate(tape) if k == "temporal")
        for idx in range(len(tape) - 1, -1, -1):
            kind, s = tape[idx]
            if kind == "gn_out":
                dx, _ = self._gn_bwd(s, dx)
            elif kind == "resnet_pop":
                dx, dskip = self._resnet_bwd(s, dx)
                dskips.append(dskip)
            elif kind == "resnet":
                if idx < first_trainable - 1:
                    break                                # nothing trainable below: stop
                dx, _ = self._resnet_bwd(s, dx)
            elif kind == "text":
                if idx < first_trainable:                # only this block's text K|V gradient is still needed
                    self._text_bwd(s, dx, stop_after_kv=True)
                    break
                dx = self._text_bwd(s, dx)
            elif kind == "temporal":
                dx = self._temporal_bwd(s, dx)
            elif kind == "push":                         # this activation was also a skip: add the gradient that came back
                dx = tops.add(dx, dskips.pop())            # LIFO: the last skip pushed is the first one consumed
            elif kind == "down":
                key, geo = s
                B, Fr, H, W = geo
                Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                dx = ops.conv3x3(tops.zero_insert2x(dx, B * Fr, Ho, Wo), self._convT(key), B * Fr, 2 * Ho, 2 * Wo)
            elif kind == "up":
                key, geo = s
                B, Fr, H, W = geo
                dx = tops.sumpool2x(ops.conv3x3(dx, self._convT(key), B * Fr, 2 * H, 2 * W), B * Fr, H, W)
        # d context = sum over the text blocks of [dK | dV] Wkv: one GEMM over the concatenated columns
        WT = self._wT.get("__kv_all__")
        if WT is None:
            WT = torch.cat([self._frozenT(tb + ".attn2.kv") for tb in self._kv_cols], 1).contiguous()      # [Dc, sum 2C]
            self._wT["__kv_all__"] = WT
        self._flush_dw()                                 # the UNet's gradient segment is final when phase A ends
        return ops.gemm(self._dkv_all, WT)

    # ================================================================================================ FSText
    def _fstext_rot(self, p, Fr, l):
        key = (p, Fr, l)
        t = self._fs_rot.get(key)
        if t is None:
            t = self.ops.rotary_table(self.wf[p + ".attn1.freqs"], Fr).repeat_interleave(l, dim=0).contiguous()
            self._fs_rot[key] = t
        return t

    def _pos_frames(self, Fr: int) -> torch.Tensor:
        """source frame of pos_embed for every output frame (F.interpolate nearest over the frame axis, fstext.py)"""
        F0 = self.pf.shapes["pos_embed"][1]
        if F0 == Fr:
            return torch.arange(Fr)
        return torch.floor(torch.arange(Fr, dtype=torch.float32) * (F0 / Fr)).long().clamp_(max=F0 - 1)

    def _fstext_fwd(self, context: torch.Tensor):
        ops, tops, W, P = self.ops, self.tops, self.wf, self.pf
        fs = self.fstext
        b, l, cdim = context.shape
        Fr, C, heads = fs.num_frames, fs.channels, fs.n_heads
        d = C // heads
        rot_dim = min(32, d)
        ctx = context.reshape(b * l, cdim)
        ctx = ops.cast_bf16(ctx.float().contiguous()) if ctx.dtype != bf16 else ctx.contiguous()
        src = self._pos_src.get(Fr)
        if src is None:
            src = self._pos_frames(Fr).to(context.device)
            self._pos_src[Fr] = src
        pos = W["pos_embed"][0, src, :l, :]                                   # fp32 master [F, l, C]
        x = (W["learnable_query"].reshape(1, 1, C) + pos).unsqueeze(0).expand(b, Fr, l, C).reshape(b * Fr * l, C)
        x = ops.cast_bf16(x.contiguous())
        tape = []
        ffn = lambda p: (p + ".norm3.w", p + ".norm3.b", p + ".ff1.w", p + ".ff1.b", p + ".ff2.w", p + ".ff2.b")
        for n in range(fs.num_layers):
            p = f"trf_blocks.{n}.transformer_blocks.0"
            n1 = ops.layernorm(x, W[p + ".norm1.w"], W[p + ".norm1.b"])
            qkv = ops.gemm(n1, W[p + ".attn1.qkv"])
            a1 = torch.empty_like(x)
            kw1 = dict(batch=b * Fr, heads=heads, head_dim=d, Sq=l, Sk=l)
            lse1 = tops.attn_lse_buffer(b * Fr, heads, l, x.device)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a1, lse=lse1, **kw1)
            x1 = ops.gemm(a1, W[p + ".attn1.out.w"], bias=W[p + ".attn1.out.b"], residual=x)
            n2 = ops.layernorm(x1, W[p + ".norm2.w"], W[p + ".norm2.b"])
            q2 = ops.gemm(n2, W[p + ".attn2.q"])
            kv = ops.gemm(ctx, W[p + ".attn2.kv"])
            a2 = torch.empty_like(x)
            kw2 = dict(batch=b, heads=heads, head_dim=d, Sq=Fr * l, Sk=l)
            lse2 = tops.attn_lse_buffer(b, heads, Fr * l, x.device)
            ops.attention(q2, kv[:, :C], kv[:, C:], a2, lse=lse2, **kw2)
            x2 = ops.gemm(a2, W[p + ".attn2.out.w"], bias=W[p + ".attn2.out.b"], residual=x1)
            x3, sff0 = self._ff_fwd(P, W, ffn(p), x2)
            tape.append((0, p, x, n1, qkv, a1, lse1, kw1, x1, n2, q2, kv, a2, lse2, kw2, sff0))
            p = f"trf_blocks.{n}.transformer_blocks.1"
            m1 = ops.layernorm(x3, W[p + ".norm1.w"], W[p + ".norm1.b"])
            cs = self._fstext_rot(p, Fr, l)
            qkv_t = ops.gemm(m1, W[p + ".attn1.qkv"], rotary=(cs, Fr * l, 0, d, rot_dim, 2 * C))
            at = torch.empty_like(x)
            kwt = dict(batch=l, heads=heads, head_dim=d, Sq=Fr, Sk=Fr, causal=True, seq_stride_rows=l, batch_stride_rows=1)
            lset = []
            for b0 in range(b):
                sl = slice(b0 * Fr * l, (b0 + 1) * Fr * l)
                ls = tops.attn_lse_buffer(l, heads, Fr, x.device)
                ops.attention(qkv_t[sl, :C], qkv_t[sl, C:2 * C], qkv_t[sl, 2 * C:], at[sl], lse=ls, **kwt)
                lset.append(ls)
            x4 = ops.gemm(at, W[p + ".attn1.out.w"], bias=W[p + ".attn1.out.b"], residual=x3)
            x5, sff1 = self._ff_fwd(P, W, ffn(p), x4)
            tape.append((1, p, x3, m1, qkv_t, at, lset, kwt, cs, sff1))
            x = x5
        y = ops.layernorm(x, W["norm.w"], W["norm.b"])
        return y, (tape, x, ctx, (b, Fr, l, C, heads, d, rot_dim, src))

    def _fstext_bwd(self, saved, dy):
        ops, tops, W, P = self.ops, self.tops, self.wf, self.pf
        tape, x_last, ctx, (b, Fr, l, C, heads, d, rot_dim, src) = saved
        G = lambda k: P.view(P.g, k)
        ffn = lambda p: (p + ".norm3.w", p + ".norm3.b", p + ".ff1.w", p + ".ff1.b", p + ".ff2.w", p + ".ff2.b")
        dx = tops.layernorm_bwd(x_last, dy, W["norm.w"], dgamma=G("norm.w"), dbeta=G("norm.b"), defer=self._cf)
        for entry in reversed(tape):
            if entry[0] == 1:
                _, p, x3, m1, qkv_t, at, lset, kwt, cs, sff1 = entry
                dx4 = self._ff_bwd(P, W, ffn(p), sff1, dx)
                dat = self._lin_bwd(P, W, p + ".attn1.out.w", p + ".attn1.out.b", at, dx4)
                dqkv = torch.empty_like(qkv_t)
                for b0 in range(b):
                    sl = slice(b0 * Fr * l, (b0 + 1) * Fr * l)
                    tops.attention_bwd(qkv_t[sl, :C], qkv_t[sl, C:2 * C], qkv_t[sl, 2 * C:], at[sl], lset[b0], dat[sl],
                                       dqkv[sl, :C], dqkv[sl, C:2 * C], dqkv[sl, 2 * C:], **kwt)
                ops.rotary_inplace(dqkv, 0, C, heads, d, rot_dim, Fr * l, self._conj(cs))
                dm1 = self._lin_bwd(P, W, p + ".attn1.qkv", None, m1, dqkv)
                dx = tops.layernorm_bwd(x3, dm1, W[p + ".norm1.w"], dres=dx4, dgamma=G(p + ".norm1.w"), dbeta=G(p + ".norm1.b"), defer=self._cf)
            else:
                _, p, x0, n1, qkv, a1, lse1, kw1, x1, n2, q2, kv, a2, lse2, kw2, sff0 = entry
                dx2 = self._ff_bwd(P, W, ffn(p), sff0, dx)
                da2 = self._lin_bwd(P, W, p + ".attn2.out.w", p + ".attn2.out.b", a2, dx2)
                dq2 = torch.empty_like(q2)
                dkv = torch.empty_like(kv)
                tops.attention_bwd(q2, kv[:, :C], kv[:, C:], a2, lse2, da2, dq2, dkv[:, :C], dkv[:, C:], **kw2)
                self._lin_bwd(P, W, p + ".attn2.kv", None, ctx, dkv, need_dx=False)       # CLIP is frozen: no d ctx
                dn2 = self._lin_bwd(P, W, p + ".attn2.q", None, n2, dq2)
                dx1 = tops.layernorm_bwd(x1, dn2, W[p + ".norm2.w"], dres=dx2, dgamma=G(p + ".norm2.w"), dbeta=G(p + ".norm2.b"), defer=self._cf)
                da1 = self._lin_bwd(P, W, p + ".attn1.out.w", p + ".attn1.out.b", a1, dx1)
                dqkv = torch.empty_like(qkv)
                tops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a1, lse1, da1, dqkv[:, :C], dqkv[:, C:2 * C],
                                   dqkv[:, 2 * C:], **kw1)
                dn1 = self._lin_bwd(P, W, p + ".attn1.qkv", None, n1, dqkv)
                dx = tops.layernorm_bwd(x0, dn1, W[p + ".norm1.w"], dres=dx1, dgamma=G(p + ".norm1.w"), dbeta=G(p + ".norm1.b"), defer=self._cf)
        self._flush_dw()
        # token embeddings: learnable_query sees every row, pos_embed[src[f], :l] the rows of frame f summed over the batch
        tops.colsum(dx, out=G("learnable_query").reshape(-1))
        gp = G("pos_embed")
        gp.zero_()
        dxf = dx.float().reshape(b, Fr, l, C).sum(0) if b > 1 else dx.float().reshape(Fr, l, C)
        if Fr <= gp.shape[1]:
            gp[0, src, :l, :] = dxf                 # every source frame is read by at most one output frame
        else:
            gp[0, :, :l, :].index_add_(0, src, dxf)

    # ================================================================================================ the step
    def forward_backward(self, model_input: torch.Tensor, target: torch.Tensor, timesteps: torch.Tensor,
                         text_cond_emb: torch.Tensor, cond_frames: int, use_graph: bool = False,
                         on_unet_grads=None) -> torch.Tensor:
        """model_input [b, 4, F, h, w] fp32 = cat[latents_x0, noisy_latents] (train.py:364-365); target = the noise
        [b, 4, F - cond, h, w]; text_cond_emb [b, 77, 768] (CLIP last hidden state).  Fills the gradient buffers and returns
        the loss (1-element device tensor).  use_graph: replay the forward + backward (~2.5k launches, shape-static) as
        hipGraphs -- eager, the step is bound by the host's launch rate, not by the GPU.
        The step has two phases: (A) FSTextTransformer forward, UNet forward, loss, UNet backward -- after it the UNet's
        gradient segment is final -- and (B) the FSTextTransformer backward.  `on_unet_grads()` is called between them: data
        parallel training starts the all-reduce of the UNet segment there, so it travels under phase B."""
        if not model_input.is_cuda and self.ops is hip_ops:
            raise hip_ops._lib.SeerHipError("SeerTrainer needs ROCm tensors: the HIP kernels are the only compute path")
        if use_graph and not self._graph_broken:
            return self._forward_backward_graph(model_input, target, timesteps, text_cond_emb, cond_frames, on_unet_grads)
        st = self._phase_a(model_input, target, timesteps, text_cond_emb, cond_frames)
        if on_unet_grads is not None:
            on_unet_grads()
        return self._phase_b(st)

    def _forward_backward_graph(self, model_input, target, timesteps, text_cond_emb, cond_frames, on_unet_grads=None):
        b = model_input.shape[0]
        t = timesteps if torch.is_tensor(timesteps) else torch.tensor([timesteps] * b)
        t = t.to(model_input.device, torch.int64).expand(b).contiguous()
        key = (tuple(model_input.shape), tuple(text_cond_emb.shape), cond_frames)
        g = self._graphs.get(key)
        if g is None:
            bufs = [model_input.float().clone(), target.float().clone(), t.clone(), text_cond_emb.float().clone()]
            self._phase_b(self._phase_a(*bufs, cond_frames))       # eager warm-up: builds the transposed frozen weights, tables
            torch.cuda.synchronize()
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(ga):
                    st = self._phase_a(*bufs, cond_frames)
                with torch.cuda.graph(gb, pool=ga.pool()):         # phase B reads what phase A kept: one memory pool
                    loss = self._phase_b(st)
            except Exception as e:                                 # capture refused: stay correct, run eagerly
                self._graph_broken = True
                import warnings
                warnings.warn(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); running eagerly")
                return self.forward_backward(model_input, target, timesteps, text_cond_emb, cond_frames, False, on_unet_grads)
            g = (ga, gb, bufs, loss, self.last_pred, st)
            self._graphs = {key: g}
        ga, gb, bufs, loss, pred, _ = g
        for dst, src in zip(bufs, (model_input, target, t, text_cond_emb)):
            dst.copy_(src)
        ga.replay()
        if on_unet_grads is not None:
            on_unet_grads()
        gb.replay()
        self.last_pred = pred
        return loss

    def _phase_a(self, model_input, target, timesteps, text_cond_emb, cond_frames):
        b, _, Fr, _, _ = model_input.shape
        assert self.fstext.num_frames == Fr, "fstext.set_numframe(F) first (train.py:187)"
        self._dw = []                              # (a walk that raised half way must not leave its queue to the next step)
        if self._cf is not None:
            self._cf = []
        self._wt_plan.run()
        y, fs_saved = self._fstext_fwd(text_cond_emb)                     # [b*F*l, Dc] bf16, rows (b, f, l)
        t = timesteps if torch.is_tensor(timesteps) else torch.tensor([timesteps] * b)
        t = t.to(model_input.device, torch.int64).expand(b).contiguous()
        pred, tape = self._unet_fwd(model_input.float().contiguous(), t, y, text_cond_emb.shape[1], cond_frames)
        loss, dpred = self.tops.mse_loss_grad(pred, target.float().contiguous(), cond_frames)
        dctx = self._unet_bwd(tape, dpred)
        if self.text_loss:      # loss += mse(text_seq.mean(1), text_cond_emb): its gradient joins the UNet's d context
            lt = self.tops.text_loss_grad(y, text_cond_emb.float().contiguous(), b, Fr, dctx)
            self.last_text_loss = lt
            loss = loss + lt
        self.last_pred = pred
        return fs_saved, dctx, loss

    def _phase_b(self, st):
        fs_saved, dctx, loss = st
        self._fstext_bwd(fs_saved, dctx)
        return loss

    def accumulate(self) -> bool:
        """fold the gradients of the micro-batch just back-propagated into the running mean (accelerate divides the loss by
        gradient_accumulation_steps, so the accumulated gradient is the MEAN over the micro-batches).  Returns True when the
        optimizer should step (`accelerator.sync_gradients`, train.py:383,392)."""
        self._micro += 1
        if self.accum > 1:
            first = (self._micro - 1) % self.accum == 0
            for P in (self.pu, self.pf):
                self.tops.axpby(P.acc, P.g, 1.0 / self.accum, 0.0 if first else 1.0)
        return self._micro % self.accum == 0

    def start_unet_allreduce(self) -> None:
        """the `on_unet_grads` hook of data-parallel training without gradient accumulation: the UNet segment (0.9 GB at full
        size) starts its all-reduce as soon as the UNet backward has finished and travels under the FSTextTransformer backward"""
        if self.pg is not None and self.accum == 1:
            import torch.distributed as dist
            self._early_hu = dist.all_reduce(self.pu.g, group=self.pg, async_op=True)

    def optimizer_step(self, lr: Optional[float] = None) -> None:
        lr = self.lr if lr is None else lr
        gu = self.pu.acc if self.accum > 1 else self.pu.g
        gf = self.pf.acc if self.accum > 1 else self.pf.g
        ws, hu, hf = 1, None, None
        if self.pg is not None:              # DDP: average the flat gradient buffers; both all-reduces are in flight at once and
            import torch.distributed as dist  # the UNet segment's clip + AdamW run under the FSTextTransformer segment's
            ws = dist.get_world_size(self.pg)
            hu, self._early_hu = self._early_hu, None          # already travelling since the end of the UNet backward?
            if hu is None:
                hu = dist.all_reduce(gu, group=self.pg, async_op=True)
            hf = dist.all_reduce(gf, group=self.pg, async_op=True)
        self.step_count += 1
        kw = dict(lr=lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, step=self.step_count)
        if hu is not None:
            hu.wait()
            self.tops.axpby(gu, gu, 1.0 / ws, 0.0)
        ss = self.tops.sumsq(gu)                                           # clip_grad_norm_(sunet.parameters()) only
        self.tops.adamw_step(self.pu.p, gu, self.pu.m, self.pu.v, grad_sumsq=ss, max_norm=self.max_grad_norm,
                             p_bf16=self.pu.pb, **kw)
        if hf is not None:
            hf.wait()
            self.tops.axpby(gf, gf, 1.0 / ws, 0.0)
        self.tops.adamw_step(self.pf.p, gf, self.pf.m, self.pf.v, p_bf16=self.pf.pb, **kw)
        self.grad_norm_sq = ss

    def train_step(self, latents_x0, latents, noise, timesteps, text_cond_emb, alphas_cumprod, lr: Optional[float] = None,
                   use_graph: bool = False):
        """train.py:355-387 after the VAE encode: DDPM add_noise, concat the conditioning latents, forward, loss, backward,
        clip, AdamW (the optimizer runs every `gradient_accumulation_steps` calls).  alphas_cumprod: the scheduler's table
        (fp32 [T])."""
        a = alphas_cumprod.to(latents.device, f32)[timesteps].reshape(-1, 1, 1, 1, 1)
        noisy = a.sqrt() * latents + (1 - a).sqrt() * noise               # DDPMScheduler.add_noise (input preparation)
        x = torch.cat([latents_x0, noisy], 2)
        loss = self.forward_backward(x, noise, timesteps, text_cond_emb, latents_x0.shape[2], use_graph=use_graph,
                                     on_unet_grads=self.start_unet_allreduce if self.pg is not None else None)
        if self.accumulate():
            self.optimizer_step(lr)
        return loss

    # ================================================================================================ checkpoints
    @torch.no_grad()
    def sync_modules(self) -> None:
        """copy the fp32 master parameters back into the `SeerUNet` / `FSTextTransformer` modules (their `state_dict()` is what
        `accelerator.save_state` serialises, train.py:395-399) and drop their packed inference weights, so that the same
        objects sample with the fine-tuned weights."""
        new = self.trainable_state_dict()
        for mod, sd in ((self.unet, new["unet"]), (self.fstext, new["fstext"])):
            params = dict(mod.named_parameters())
            for k, v in sd.items():
                params[k].copy_(v.reshape(params[k].shape))
        for mod in (self.unet, self.fstext):          # packed bf16 copies are rebuilt on the next forward
            if hasattr(mod, "_engine"):
                mod._engine = None
            if hasattr(mod, "_invalidate"):
                mod._invalidate()

    @torch.no_grad()
    def reload_from_modules(self) -> None:
        """re-read the trainable tensors from the modules (after a `load_state_dict`: resume, train.py:268-272) into the flat
        master buffers IN PLACE: the working weights and the captured graphs are views of / read those buffers by address"""
        packed_u = _pack_temporal_fp32(dict(self.unet.state_dict()))
        packed_f = _pack_fstext_fp32(dict(self.fstext.state_dict()), self.fstext.num_layers)
        for P, packed in ((self.pu, packed_u), (self.pf, packed_f)):
            for k in P.names:
                P.view(P.p, k).copy_(packed[k].to(self.device, f32).reshape(P.shapes[k]))
            P.pb.copy_(P.p)
        for k in list(self._wT):                      # transposed copies of trainable matrices are stale now
            if k in self.trainable_u or k in self.wf:
                del self._wT[k]

    def save_state(self, save_path: str, global_step: Optional[int] = None, epoch: int = 0) -> str:
        """the files of `accelerator.save_state(save_path)` that inference reads back (inference_img.py:98-104):
        `pytorch_model.bin` (SeerUNet) and `pytorch_model_1.bin` (FSTextTransformer), plus the Adam state of this trainer
        (`optimizer.bin`: flat packed buffers) and the step counters of train.py:398."""
        import os
        os.makedirs(save_path, exist_ok=True)
        self.sync_modules()
        cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}
        torch.save(cpu(self.unet.state_dict()), os.path.join(save_path, "pytorch_model.bin"))
        torch.save(cpu(self.fstext.state_dict()), os.path.join(save_path, "pytorch_model_1.bin"))
        torch.save({"step_count": self.step_count, "micro": self._micro, "epoch": epoch, "global_step": global_step,
                    "unet": {"m": self.pu.m.cpu(), "v": self.pu.v.cpu()}, "fstext": {"m": self.pf.m.cpu(), "v": self.pf.v.cpu()}},
                   os.path.join(save_path, "optimizer.bin"))
        return save_path

    def load_optimizer_state(self, save_path: str) -> None:
        import os
        st = torch.load(os.path.join(save_path, "optimizer.bin"), map_location="cpu")
        self.step_count, self._micro = int(st["step_count"]), int(st["micro"])
        for P, key in ((self.pu, "unet"), (self.pf, "fstext")):
            P.m.copy_(st[key]["m"])
            P.v.copy_(st[key]["v"])

    def trainable_state_dict(self) -> "Dict[str, Dict[str, torch.Tensor]]":
        """{'unet': {...}, 'fstext': {...}} fp32 tensors under the REFERENCE's parameter names (unpacked)."""
        return self.trainable_state_dict_of(self.pu.p, self.pf.p)

    def trainable_state_dict_of(self, flat_u: torch.Tensor, flat_f: torch.Tensor) -> "Dict[str, Dict[str, torch.Tensor]]":
        """the same unpacking applied to any pair of flat buffers (parameters, gradients, Adam moments)"""
        out_u: Dict[str, torch.Tensor] = {}
        for k in self.pu.names:
            v = self.pu.view(flat_u, k).detach().clone()
            if k.endswith(".qkv"):
                p = k[: -len(".qkv")]
                q, kk, vv = v.chunk(3, 0)
                out_u[p + ".to_q.weight"], out_u[p + ".to_k.weight"], out_u[p + ".to_v.weight"] = q, kk, vv
            elif k.endswith("ff.net.0.proj.weight") or k.endswith("ff.net.0.proj.bias"):
                inv = torch.empty_like(geglu_row_order(v.shape[0] // 2))
                order = geglu_row_order(v.shape[0] // 2)
                inv[order] = torch.arange(order.numel())
                out_u[k] = v[inv.to(v.device)]
            elif k.endswith((".proj_in.weight", ".proj_out.weight")):
                out_u[k] = v.reshape(v.shape[0], v.shape[1], 1, 1)
            else:
                out_u[k] = v
        out_f: Dict[str, torch.Tensor] = {}
        for k in self.pf.names:
            v = self.pf.view(flat_f, k).detach().clone()
            if k.endswith(".attn1.qkv"):
                p = k[: -len(".qkv")]
                q, kk, vv = v.chunk(3, 0)
                out_f[p + ".to_q.weight"], out_f[p + ".to_k.weight"], out_f[p + ".to_v.weight"] = q, kk, vv
            elif k.endswith(".attn2.q"):
                out_f[k[:-2] + ".to_q.weight"] = v
            elif k.endswith(".attn2.kv"):
                kk, vv = v.chunk(2, 0)
                out_f[k[:-3] + ".to_k.weight"], out_f[k[:-3] + ".to_v.weight"] = kk, vv
            elif k.endswith(".out.w"):
                out_f[k[:-6] + ".to_out.0.weight"] = v
            elif k.endswith(".out.b"):
                out_f[k[:-6] + ".to_out.0.bias"] = v
            elif k.endswith((".ff1.w", ".ff1.b")):
                order = geglu_row_order(v.shape[0] // 2)
                inv = torch.empty_like(order)
                inv[order] = torch.arange(order.numel())
                out_f[k[:-6] + ".ff.net.0.proj." + ("weight" if k.endswith("w") else "bias")] = v[inv.to(v.device)]
            elif k.endswith((".ff2.w", ".ff2.b")):
                out_f[k[:-6] + ".ff.net.2." + ("weight" if k.endswith("w") else "bias")] = v
            elif k.endswith((".w", ".b")) and ".norm" in k or k in ("norm.w", "norm.b"):
                out_f[k[:-2] + (".weight" if k.endswith(".w") else ".bias")] = v
            else:
                out_f[k] = v
        return {"unet": out_u, "fstext": out_f}
