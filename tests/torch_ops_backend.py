"""TEST INFRASTRUCTURE ONLY -- a plain-torch (CPU) stand-in for `seervideoldm_amd.ops` with the SAME call signatures
and the same storage rounding (bf16 tensors in/out, fp32 math inside).

It lets the CPU test-suite exercise the HOST logic of the product (weight packing, kernel schedule of `_Engine`, skip /
concat wiring, window and rotary parameters, frame sharding and its collectives over gloo) against the oracle without a
GPU.  It is never imported by the product package; the product's only backend is libseer_hip.so.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

bf16 = torch.bfloat16

# host-side containers of the product (pure torch, no kernel behind them): the accumulated fixed-point column sums and their arena
from seervideoldm_amd.ops import ColSumsFx, FxArena, groupnorm_stats_from_fx  # noqa: E402,F401

# EXACT = True: every contraction accumulates in float64 (then rounds once to fp32), so a row's result does not depend on how many
# OTHER rows the call holds -- what the sharded-vs-unsharded tests need to compare bit for bit (an fp32 BLAS blocks by shape)
EXACT = False
FX_SCALE = float(1 << 20)


def _mm64(a, bt):
    """[..., K] @ [K, N] in float64, one row at a time (a batched product of 1 x K rows): a float64 BLAS blocks by shape too -- a row
    of `a @ bt` differs in its last bits with the number of rows beside it (measured: 2 of 6 row subsets of a 512-row product), and
    once in many runs the difference crosses an fp32 rounding boundary and travels through the network"""
    lead = a.shape[:-1]
    a2 = a.double().reshape(-1, 1, a.shape[-1])
    return torch.bmm(a2, bt.double().unsqueeze(0).expand(a2.shape[0], -1, -1)).reshape(*lead, bt.shape[1])


def _mm(a, bt):
    return _mm64(a, bt).float() if EXACT else a @ bt


def _fx_sums(res, colsum_batch):
    """ColSumsFx of a stored output for colsum_batch = (B, arena): every element rounded on its own, as
    seer_groupnorm_stats_fx does (the MFMA producers round per tile partial: the same statistics to 2^-21 per partial)"""
    if not isinstance(colsum_batch, tuple):
        return None
    B, arena = colsum_batch
    v = res.float().reshape(B, -1, res.shape[1])
    buf = arena.take(1, B, res.shape[1]) if arena is not None else None
    if buf is None:
        buf = torch.zeros((1, B, 2, res.shape[1]), dtype=torch.int64)
    buf[0, :, 0] += torch.round(v * FX_SCALE).to(torch.int64).sum(1)
    buf[0, :, 1] += torch.round(v * v * FX_SCALE).to(torch.int64).sum(1)
    return ColSumsFx(buf, res.shape[1])


def _deinterleave_geglu(acc):
    M, N = acc.shape
    a = acc.reshape(M, N // 32, 2, 16)
    return a[:, :, 0].reshape(M, N // 2), a[:, :, 1].reshape(M, N // 2)


def _epilogue(acc, bias, geglu, rowvec, rows_per_batch, silu, residual, out_f32, out):
    if bias is not None:
        acc = acc + bias
    if geglu:
        val, gate = _deinterleave_geglu(acc)
        acc = val * F.gelu(gate)
    if rowvec is not None:
        acc = acc + rowvec.repeat_interleave(rows_per_batch, 0)[: acc.shape[0]]
    if silu:
        acc = F.silu(acc)
    if residual is not None:
        acc = acc + residual.float()
    res = acc if (out_f32 or (out is not None and out.dtype == torch.float32)) else acc.to(bf16)
    if out is not None:
        out.copy_(res)
        return out
    return res


class RowStats:
    """stand-in of ops.RowStats: (sum, sum of squares) per row of a GEMM output, here as fp64"""
    def __init__(self, tot):
        self.tot = tot


def fold_layernorm(w, gamma, beta, bias=None):
    """the product's own host-side folding (pure torch): the thing under test on CPU"""
    from seervideoldm_amd import ops as product_ops
    return product_ops.fold_layernorm(w, gamma, beta, bias)


def gemm(a, w, *, bias=None, residual=None, rowvec=None, rows_per_batch=0, a2=None, geglu=False, silu=False,
         out_f32=False, out=None, tile=0, splits=0, rotary=None, col_scale=None, colsum_batch=0, rowstat=False, ln=None):
    A = a.float() if a2 is None else torch.cat([a.float(), a2.float()], 1)
    assert A.shape[1] % 64 == 0 and w.dtype == bf16 and a.dtype == bf16
    acc = _mm(A, w.float().t())
    if ln is not None:
        # seer_gemm_desc::ln_rowstat: rstd * (x W'^T - mean * wsum), the term the kernel applies before everything else
        rs, wsum, eps = ln
        K = A.shape[1]
        mean = (rs.tot[:, 0] / K).float()
        var = (rs.tot[:, 1] / K).float() - mean * mean
        rstd = torch.rsqrt(var.clamp_min(0) + eps)
        acc = acc * rstd[:, None] - (mean * rstd)[:, None] * wsum[None, :]
    res = _gemm_tail(acc, bias, residual, rowvec, rows_per_batch, geglu, silu, out_f32, out, rotary, col_scale)
    res.colsums = _fx_sums(res, colsum_batch)
    if rowstat:
        v = res.double()
        res.rowstats = RowStats(torch.stack([v.sum(1), (v * v).sum(1)], 1))
    elif hasattr(res, "rowstats"):
        res.rowstats = None
    return res


def _gemm_tail(acc, bias, residual, rowvec, rows_per_batch, geglu, silu, out_f32, out, rotary, col_scale):
    if (rotary is not None or col_scale is not None) and bias is not None:
        acc = acc + bias                # the kernel's order: bias, (GEGLU, row vector,) rotary, column scale, residual
        bias = None
    if rotary is not None:
        table, tpb, pos_off, hd, rd, cols = rotary
        assert bias is None and residual is None and not geglu
        rows = acc.shape[0]
        pos = (torch.arange(rows) % tpb) + pos_off
        c, s = table[pos, :, 0], table[pos, :, 1]
        t = acc[:, :cols].reshape(rows, cols // hd, hd)
        x0, x1 = t[..., :rd:2], t[..., 1:rd:2]
        rot = torch.stack([x0 * c[:, None] - x1 * s[:, None], x1 * c[:, None] + x0 * s[:, None]], -1).flatten(-2)
        acc = torch.cat([torch.cat([rot, t[..., rd:]], -1).reshape(rows, cols), acc[:, cols:]], 1)
    if col_scale is not None:
        assert bias is None and residual is None and not geglu
        acc = torch.cat([acc[:, :col_scale[1]] * col_scale[0], acc[:, col_scale[1]:]], 1)
    return _epilogue(acc, bias, geglu, rowvec, rows_per_batch, silu, residual, out_f32, out)


# ---- stand-in of ops.rowchain (seer_rowchain_c320): the row-local chains in front of the attention launches, one call -----------------
ROWCHAIN_C, ROWCHAIN_ROWS = 320, 96
ROWCHAIN_MIN_ROWS = 96         # the CPU tests' small shapes take the chain too (the product asks the device: ops.rowchain_pays)


def rowchain_pack(w):
    """the kernel reads its matrices in a fragment order; the stand-in keeps [n * 320, 320] (tagged: only rowchain() may read it)"""
    assert w.dtype == bf16 and w.shape[1] == ROWCHAIN_C and w.shape[0] % ROWCHAIN_C == 0
    p = w.clone()
    p.rowchain_packed = True
    return p


def rowchain_pays(rows, n_cu=None, products=4):
    return rows >= ROWCHAIN_MIN_ROWS


def rowchain(inp, w1f, *, b1=None, gn=None, res=None, h_out=True, ln=None, w2f=None, out=None, col_scale=None, rotary=None):
    assert getattr(w1f, "rowchain_packed", False) and inp.shape[1] == ROWCHAIN_C and inp.dtype == bf16
    M = inp.shape[0]
    x = inp.float()
    if gn is not None:
        stats, count, eps, gamma, beta, rows_pb = gn[:6]
        if rows_pb < ROWCHAIN_ROWS:
            return None                                                   # SEER_ENOSYS: a tile would span three batch elements
        B = M // rows_pb
        if isinstance(stats, ColSumsFx):
            G = gn[6]
            g = stats.buf.sum(0).reshape(B, 2, G, ROWCHAIN_C // G).sum(3).double() / FX_SCALE
            mean, ex2 = (g[:, 0] / count).float(), (g[:, 1] / count)
            var = (ex2 - (g[:, 0] / count) ** 2).clamp_min(0).float()
        else:
            G = stats.shape[1]
            mean = stats[..., 0] / count
            var = (stats[..., 1] / count - mean * mean).clamp_min(0)
        sc = (torch.rsqrt(var + eps)[:, :, None] * gamma.reshape(G, -1)[None]).reshape(B, ROWCHAIN_C)        # per (batch element, channel)
        sh = beta[None] - (mean[:, :, None].expand(-1, -1, ROWCHAIN_C // G).reshape(B, ROWCHAIN_C)) * sc
        x = (x.reshape(B, rows_pb, ROWCHAIN_C) * sc[:, None] + sh[:, None]).reshape(M, ROWCHAIN_C).to(bf16).float()   # the tile is 16-bit
    acc = _mm(x, w1f.float().t())
    if b1 is not None:
        acc = acc + b1
    if res is not None:
        acc = acc + res.float()
    hq = acc.to(bf16)
    h = None
    if h_out is not False:
        if h_out is True:
            h = hq
        else:
            h_out.copy_(hq)
            h = h_out
        h.colsums = None
        h.rowstats = None
    o = None
    if w2f is not None:
        assert getattr(w2f, "rowchain_packed", False)
        n2 = w2f.shape[0] // ROWCHAIN_C
        t = hq.float()
        if ln is not None:
            t = F.layer_norm(t, (ROWCHAIN_C,), ln[0], ln[1], ln[2]).to(bf16).float()
        acc2 = _mm(t, w2f.float().t())
        rot = None if rotary is None else (rotary[0], rotary[1], rotary[2], rotary[3], rotary[4], rotary[5] * ROWCHAIN_C)
        cs = None if col_scale is None else (col_scale[0], col_scale[1] * ROWCHAIN_C)
        o = _gemm_tail(acc2, None, None, None, 0, False, False, False, out, rot, cs)
    return h, o


def gemm_batched(a, w, *, trans_out=False, out=None, bias=None, out_f32=False, tile=0, col_scale=None):
    acc = torch.einsum("bmk,bnk->bmn" if w.dim() == 3 else "bmk,nk->bmn", a.float(), w.float())
    if bias is not None:
        acc = acc + bias
    if col_scale is not None:
        acc[..., :col_scale[1]] = acc[..., :col_scale[1]] * col_scale[0]
    if trans_out:
        acc = acc.transpose(1, 2).contiguous()
    return acc if out_f32 else acc.to(bf16)


def conv3x3(x, w, n_img, Hin, Win, *, stride=1, upsample=False, bias=None, residual=None, rowvec=None,
            rows_per_batch=0, out=None, tile=0, splits=0, colsum_batch=0):
    Ci, Co = x.shape[1], w.shape[0]
    assert Ci % 64 == 0 and w.shape[1] == 9 * Ci
    xi = x.float().reshape(n_img, Hin, Win, Ci).permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    wt = w.float().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    if EXACT:       # one image per call: the same shape whatever the number of images (frames) the rank holds
        wd = wt.double()
        y = torch.cat([F.conv2d(xi[i:i + 1].double(), wd, None, stride=stride, padding=1) for i in range(xi.shape[0])], 0)
        y = y.float().permute(0, 2, 3, 1).reshape(-1, Co)
    else:
        y = F.conv2d(xi, wt, None, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    res = _epilogue(y, bias, False, rowvec, rows_per_batch, False, residual, False, out)
    res.colsums = _fx_sums(res, colsum_batch)
    return res


def conv_up2x(x, w4, n_img, Hin, Win, *, bias=None, out=None, tile=0, colsum_batch=0):
    """the four 2x2 phase convs of weights.pack_conv3x3_up_phases, written out: phase (a, b) reads source rows y+a-1, y+a"""
    Ci, Co = x.shape[1], w4.shape[1]
    xi = F.pad(x.float().reshape(n_img, Hin, Win, Ci), (0, 0, 1, 1, 1, 1))          # zero border of one source pixel
    y = torch.zeros((n_img, 2 * Hin, 2 * Win, Co))
    wf = w4.float().reshape(4, Co, 2, 2, Ci)
    for a in range(2):
        for b in range(2):
            acc = 0
            for ty in range(2):
                for tx in range(2):
                    src = xi[:, a + ty:a + ty + Hin, b + tx:b + tx + Win, :]        # source (y + a - 1 + ty, x + b - 1 + tx)
                    acc = acc + (_mm64(src, wf[a * 2 + b, :, ty, tx, :].t()) if EXACT else src @ wf[a * 2 + b, :, ty, tx, :].t())
            y[:, a::2, b::2, :] = acc.float() if EXACT else acc
    res = _epilogue(y.reshape(-1, Co), bias, False, None, 0, False, None, False, out)
    res.colsums = _fx_sums(res, colsum_batch)
    return res


LOG2E = 1.4426950408889634


def qk_prescale(head_dim, scale=None):
    return (head_dim ** -0.5 if scale is None else scale) * LOG2E


def _tok_index(batch, ws, Fr, H, W):
    """token index [nW, Fr*ws*ws] of every window position, order (f, wy, wx); windows ordered (wy_blk, wx_blk)"""
    f = torch.arange(Fr)[:, None, None]
    wy = torch.arange(ws)[None, :, None]
    wx = torch.arange(ws)[None, None, :]
    idx = []
    for by in range(H // ws):
        for bx in range(W // ws):
            idx.append((f * H * W + (by * ws + wy) * W + bx * ws + wx).reshape(-1))
    return torch.stack(idx)


def attention(q, k, v, out, *, batch, heads, head_dim, Sq, Sk, causal=False, scale=None, window=None, Fq=None,
              causal_offset=0, seq_stride_rows=1, batch_stride_rows=None, lse=None, q_prescaled=False, variant=0):
    assert head_dim in (40, 80, 96, 160), "the flash kernels are built for head_dim 40/80/96/160"
    C = heads * head_dim
    scale = head_dim ** -0.5 if scale is None else scale
    if q_prescaled:         # q carries scale * log2(e): softmax_e(scale * qk) == softmax_2(q'k)
        scale = 0.6931471805599453
    if batch_stride_rows is not None:
        # row(b, s) = b*batch_stride_rows + s*seq_stride_rows: gather to the contiguous form, run, scatter back
        assert window is None
        rq = (torch.arange(batch)[:, None] * batch_stride_rows + torch.arange(Sq)[None] * seq_stride_rows).reshape(-1)
        rk = (torch.arange(batch)[:, None] * batch_stride_rows + torch.arange(Sk)[None] * seq_stride_rows).reshape(-1)
        tmp = torch.empty((batch * Sq, C), dtype=bf16)
        attention(q[rq], k[rk], v[rk], tmp, batch=batch, heads=heads, head_dim=head_dim, Sq=Sq, Sk=Sk, causal=causal,
                  scale=scale, causal_offset=causal_offset)
        out[rq, :C] = tmp
        return out
    if window is None:
        qq = q[:, :C].float().reshape(batch, Sq, heads, head_dim).permute(0, 2, 1, 3)
        kk = k[:, :C].float().reshape(batch, Sk, heads, head_dim).permute(0, 2, 1, 3)
        vv = v[:, :C].float().reshape(batch, Sk, heads, head_dim).permute(0, 2, 1, 3)
    else:
        ws, Fr, H, W = window
        Fq = Fr if Fq is None else Fq
        assert Sq == Fq * ws * ws and Sk == Fr * ws * ws
        iq, ik = _tok_index(batch, ws, Fq, H, W), _tok_index(batch, ws, Fr, H, W)      # [nW, S]
        gat = lambda t, idx, Ft: t[:, :C].float().reshape(batch, Ft * H * W, heads, head_dim)[:, idx]  # [B,nW,S,h,d]
        qq = gat(q, iq, Fq).permute(1, 0, 3, 2, 4).reshape(-1, heads, Sq, head_dim)
        kk = gat(k, ik, Fr).permute(1, 0, 3, 2, 4).reshape(-1, heads, Sk, head_dim)
        vv = gat(v, ik, Fr).permute(1, 0, 3, 2, 4).reshape(-1, heads, Sk, head_dim)
    if EXACT:
        qq, kk, vv = qq.double(), kk.double(), vv.double()
    s = torch.einsum("bhqd,bhkd->bhqk", qq, kk) * scale
    if causal:
        i = torch.arange(Sq)[:, None] + causal_offset
        j = torch.arange(Sk)[None, :]
        s = s.masked_fill(~(j <= i), float("-inf"))
    o = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), vv).float()      # [nb, heads, Sq, d]
    if window is None:
        res = o.permute(0, 2, 1, 3).reshape(batch * Sq, C)
    else:
        nW = iq.shape[0]
        o = o.reshape(nW, batch, heads, Sq, head_dim).permute(1, 0, 3, 2, 4)           # [B, nW, S, h, d]
        res = torch.zeros(batch, Fq * H * W, C)
        res[:, iq.reshape(-1)] = o.reshape(batch, nW * Sq, C)
        res = res.reshape(-1, C)
    out[:, :C].copy_(res.to(bf16))
    return out


def rotary_table(freqs, T):
    ang = torch.arange(T, dtype=torch.float32)[:, None] * freqs[None, :]
    return torch.stack([ang.cos(), ang.sin()], -1)


def rotary_inplace(x, col0_q, col0_k, heads, head_dim, rot_dim, tokens_per_batch, cos_sin, pos_offset=0):
    rows = x.shape[0]
    pos = (torch.arange(rows) % tokens_per_batch) + pos_offset
    c, s = cos_sin[pos, :, 0], cos_sin[pos, :, 1]                     # [rows, half]
    for col0 in (col0_q, col0_k):
        t = x[:, col0:col0 + heads * head_dim].float().reshape(rows, heads, head_dim)
        tr = t[..., :rot_dim]
        a, b = tr[..., 0::2], tr[..., 1::2]
        ra = a * c[:, None] - b * s[:, None]
        rb = b * c[:, None] + a * s[:, None]
        t = torch.cat([torch.stack([ra, rb], -1).flatten(-2), t[..., rot_dim:]], -1)
        x[:, col0:col0 + heads * head_dim] = t.reshape(rows, -1).to(bf16)


def groupnorm_stats(x1, x2, batch, groups, stats):
    xc = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    C = xc.shape[1]
    xg = xc.reshape(batch, -1, groups, C // groups)
    stats[..., 0] = xg.sum(dim=(1, 3))
    stats[..., 1] = (xg * xg).sum(dim=(1, 3))
    return stats


def groupnorm_apply(x1, x2, batch, groups, stats, count, eps, gamma, beta, silu, out=None):
    xc = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    C = xc.shape[1]
    mean = stats[..., 0] / count
    var = (stats[..., 1] / count - mean * mean).clamp_min(0)
    rstd = torch.rsqrt(var + eps)
    xg = xc.reshape(batch, -1, groups, C // groups)
    y = ((xg - mean[:, None, :, None]) * rstd[:, None, :, None]).reshape(-1, C) * gamma + beta
    if silu:
        y = F.silu(y)
    return y.to(bf16)


def groupnorm_stats_fx(x, batch, arena=None):
    """stand-in of ops.groupnorm_stats_fx (seer_groupnorm_stats_fx): exact per-element fixed-point sums from the activations"""
    return _fx_sums(x, (batch, arena))


def groupnorm_apply_fx(x1, x2, fx1, fx2, batch, groups, count, eps, gamma, beta, silu, out=None, stats_out=None):
    """stand-in of seer_groupnorm_apply_fx: replicas and a group's channels added as integers, one conversion per group (double)"""
    tot = fx1.buf.sum(0) if fx2 is None else torch.cat([fx1.buf.sum(0), fx2.buf.sum(0)], dim=2)      # [B, 2, C] int64
    C = tot.shape[2]
    g = tot.reshape(batch, 2, groups, C // groups).sum(3).double() / FX_SCALE                          # [B, 2, G]
    mean = g[:, 0] / count
    var = (g[:, 1] / count - mean * mean).clamp_min(0)
    stats = torch.stack([mean.float(), torch.rsqrt(var.float() + eps)], -1)                              # (mean, rstd) per (b, g)
    xc = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    xg = xc.reshape(batch, -1, groups, C // groups)
    y = ((xg - stats[:, None, :, None, 0]) * stats[:, None, :, None, 1]).reshape(-1, C) * gamma + beta
    if silu:
        y = F.silu(y)
    if stats_out is not None:
        stats_out.copy_(g.permute(0, 2, 1).float())
    return y.to(bf16)


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    y = F.layer_norm(x.float(), (x.shape[1],), gamma, beta, eps).to(bf16)
    if out is not None:
        out.copy_(y)
        return out
    return y


def softmax_rows(x, scale, out=None):
    return (x.float() * scale).softmax(-1).to(bf16)


def timestep_embedding(t, dim, flip_sin_to_cos, freq_shift):
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / (half - freq_shift)
    arg = t[:, None].float() * torch.exp(exponent)[None]
    return torch.cat([arg.cos(), arg.sin()], -1) if flip_sin_to_cos else torch.cat([arg.sin(), arg.cos()], -1)


def linear_smallm(x, w, bias, *, silu_in=False, silu_out=False):
    xi = F.silu(x) if silu_in else x
    y = _mm(xi, w.float().t())
    if bias is not None:
        y = y + bias
    return F.silu(y) if silu_out else y


def conv_in(x, w_khwc, bias):
    B, Cin, Fr, H, W = x.shape
    wt = w_khwc.permute(3, 2, 0, 1)
    xi = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, Cin, H, W)
    if EXACT:
        y = torch.cat([F.conv2d(xi[i:i + 1].double(), wt.double(), bias.double(), padding=1) for i in range(xi.shape[0])], 0).float()
    else:
        y = F.conv2d(xi, wt, bias, padding=1)
    return y.permute(0, 2, 3, 1).reshape(-1, y.shape[1]).to(bf16)


def conv_out(x, w_ohwc, bias, B, Fr, H, W):
    C0 = x.shape[1]
    if w_ohwc.dim() == 2:      # bf16 [Cout, 9*C0] in conv3x3 packing (the MFMA path of the HIP backend)
        w_ohwc = w_ohwc.float().reshape(w_ohwc.shape[0], 3, 3, C0)
    xi = x.float().reshape(B * Fr, H, W, C0).permute(0, 3, 1, 2)
    wt = w_ohwc.permute(0, 3, 1, 2)
    if EXACT:
        y = torch.cat([F.conv2d(xi[i:i + 1].double(), wt.double(), bias.double(), padding=1) for i in range(xi.shape[0])], 0).float()
    else:
        y = F.conv2d(xi, wt, bias, padding=1)
    return y.reshape(B, Fr, -1, H, W).permute(0, 2, 1, 3, 4).contiguous()


def cast_bf16(x):
    return x.to(bf16)
