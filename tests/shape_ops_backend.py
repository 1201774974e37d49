"""TEST INFRASTRUCTURE ONLY -- a SHAPE-ONLY stand-in for `seervideoldm_amd.ops`: same call signatures, outputs are empty tensors of
the right shape and dtype on the inputs' device (meant for torch's "meta" device: nothing is computed, nothing is allocated).

It lets a CPU test walk the kernel schedule of the FULL-SIZE `_Engine` (BASELINE config 2: 1.08 G parameters) in a second and
count launches / algorithmic FLOPs / bytes through `seervideoldm_amd.profiler.TimedOps` -- the accounting contract of bench.py's
`roofline` object.  Never imported by the product package.
"""
from __future__ import annotations

import math

import torch

bf16 = torch.bfloat16
LOG2E = 1.4426950408889634

# rows from which the library gives the GEGLU projection to the weight-stationary kernel and therefore REFUSES to fold the
# LayerNorm into it (seer_gemm_lnfold_ok, csrc/gemm.hip: the ten level-0 ff.net.0 launches of config 2)
LN_FOLD_REFUSED_GEGLU_ROWS = 16384


class RowStats:
    def __init__(self, rows):
        self.rows = rows


def fold_layernorm(w, gamma, beta, bias=None):
    N = w.shape[0]
    return (torch.empty(w.shape, dtype=bf16, device=w.device), torch.empty((N,), dtype=torch.float32, device=w.device),
            torch.empty((N,), dtype=torch.float32, device=w.device))


def gemm(a, w, *, bias=None, residual=None, rowvec=None, rows_per_batch=0, a2=None, geglu=False, silu=False, out_f32=False,
         out=None, tile=0, splits=0, rotary=None, col_scale=None, colsum_batch=0, rowstat=False, ln=None):
    M, N = a.shape[0], w.shape[0]
    assert a.shape[1] + (0 if a2 is None else a2.shape[1]) == w.shape[1]
    if ln is not None and geglu and M >= LN_FOLD_REFUSED_GEGLU_ROWS:
        return None                                         # nothing launched: the caller runs layernorm + the plain weights
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float32 if out_f32 else bf16, device=a.device)
    assert out.shape == (M, n_out)
    out.colsums = None
    out.rowstats = RowStats(M) if rowstat else None
    return out


FF_FUSED_C, FF_FUSED_ROWS, FF_FUSED_MIN_ROWS = 320, 96, 18432


def ff_fused_pays(rows, n_cu=256):
    wgs = -(-rows // FF_FUSED_ROWS)
    return rows >= FF_FUSED_MIN_ROWS and wgs / (n_cu * -(-wgs // n_cu)) >= 0.74


def ff_fused_pack(w1, wcat):
    return torch.empty_like(w1), torch.empty_like(wcat)


def ff_fused(h, x, gamma, beta, w1f, b1, wcf, bcat, *, eps=1e-5, out=None, colsum_batch=0, pre=None):
    M, Cc = h.shape
    if Cc != FF_FUSED_C or M == 0:
        return None
    out = torch.empty((M, Cc), dtype=bf16, device=h.device)
    out.colsums = None
    out.rowstats = None
    return out


ROWCHAIN_C, ROWCHAIN_ROWS = 320, 96


def rowchain_pack(w):
    return torch.empty_like(w)


def rowchain_pays(rows, n_cu=256, products=4):
    wgs = -(-rows // ROWCHAIN_ROWS)
    fill = wgs / (n_cu * -(-wgs // n_cu))
    if products >= 4 and wgs <= n_cu:
        return rows >= 12288
    return rows >= 18432 and fill >= 0.74


def rowchain(inp, w1f, *, b1=None, gn=None, res=None, h_out=True, ln=None, w2f=None, out=None, col_scale=None, rotary=None):
    M, Cc = inp.shape
    if gn is not None and gn[5] < ROWCHAIN_ROWS:
        return None
    h = None
    if h_out is not False:
        h = torch.empty((M, Cc), dtype=bf16, device=inp.device) if h_out is True else h_out
        h.colsums = None
        h.rowstats = None
    o = None if w2f is None else torch.empty((M, w2f.numel() // Cc), dtype=bf16, device=inp.device)
    return h, o


def gemm_batched(a, w, *, trans_out=False, out=None, bias=None, out_f32=False, tile=0, col_scale=None):
    Bt, M, _ = a.shape
    N = w.shape[-2]
    return torch.empty((Bt, N, M) if trans_out else (Bt, M, N), dtype=torch.float32 if out_f32 else bf16, device=a.device)


def conv3x3(x, w, n_img, Hin, Win, *, stride=1, upsample=False, bias=None, residual=None, rowvec=None, rows_per_batch=0,
            out=None, tile=0, splits=0, pad_after_only=False, colsum_batch=0):
    Hs, Ws = (2 * Hin, 2 * Win) if upsample else (Hin, Win)
    Ho, Wo = (Hs - 1) // stride + 1, (Ws - 1) // stride + 1
    assert w.shape[1] == 9 * x.shape[1] and x.shape[0] == n_img * Hin * Win
    out = torch.empty((n_img * Ho * Wo, w.shape[0]), dtype=bf16, device=x.device)
    out.colsums = None
    return out


def conv_up2x(x, w4, n_img, Hin, Win, *, bias=None, out=None, tile=0, colsum_batch=0):
    assert w4.shape[0] == 4 and w4.shape[2] == 4 * x.shape[1]
    out = torch.empty((n_img * 4 * Hin * Win, w4.shape[1]), dtype=bf16, device=x.device)
    out.colsums = None
    return out


def qk_prescale(head_dim, scale=None):
    return (scale if scale is not None else 1.0 / math.sqrt(head_dim)) * LOG2E


def attention(q, k, v, out, **kw):
    return out


def rotary_table(freqs, T):
    return torch.empty((T, freqs.shape[0], 2), dtype=torch.float32, device=freqs.device)


def groupnorm_stats(x1, x2, batch, groups, stats):
    return stats


def groupnorm_apply(x1, x2, batch, groups, stats, count, eps, gamma, beta, silu, out=None):
    C = x1.shape[1] + (0 if x2 is None else x2.shape[1])
    return torch.empty((x1.shape[0], C), dtype=bf16, device=x1.device)


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    return torch.empty(x.shape, dtype=bf16, device=x.device)


def timestep_embedding(t, dim, flip_sin_to_cos, freq_shift):
    return torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)


def linear_smallm(x, w, bias, *, silu_in=False, silu_out=False):
    return torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)


def conv_in(x, w_khwc, bias):
    B, _, Fr, H, W = x.shape
    return torch.empty((B * Fr * H * W, w_khwc.shape[-1]), dtype=bf16, device=x.device)


def conv_out(x, w, bias, B, Fr, H, W):
    return torch.empty((B, 4, Fr, H, W), dtype=torch.float32, device=x.device)


def cast_bf16(x):
    return torch.empty(x.shape, dtype=bf16, device=x.device)
