"""Tensor-level wrappers over the training entry points of the C ABI (include/seer_hip.h, "training step").

Same rules as ops.py: torch supplies device memory and the stream, libseer_hip.so does the arithmetic, there is no CPU
path.  The matrix products of the backward pass are plain ops.gemm / ops.conv3x3 calls on transposed / repacked operands
(see the header); this module holds the rest.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import AttnBwdDesc, check
from .ops import _p, _req, _stream, bf16


def attn_lse_buffer(batch: int, heads: int, Sq: int, device, window=None) -> torch.Tensor:
    nb = batch if window is None else batch * (window[2] // window[0]) * (window[3] // window[0])
    return torch.empty((nb * heads, Sq), device=device, dtype=torch.float32)


def attention_bwd(q, k, v, out, lse, dout, dq, dk, dv, **kw) -> None:
    """Backward of ops.attention(q, k, v, out, lse=lse, **kw): dq/dk/dv are 2-D views addressed like q/k/v (row stride =
    token stride), e.g. the column slices of one [tokens, 3C] gradient buffer."""
    fwd = ops.attention(q, k, v, out, lse=lse, _desc_only=True, **kw)
    for t, n in ((dout, "dout"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _req(t, bf16, n)
        assert t.dim() == 2 and t.stride(1) == 1
    assert dout.shape[0] == out.shape[0] and dq.shape[0] == q.shape[0] and dk.shape[0] == k.shape[0] and dv.shape[0] == v.shape[0]
    d = AttnBwdDesc()
    d.fwd = fwd
    d.dO, d.dQ, d.dK, d.dV = _p(dout), _p(dq), _p(dk), _p(dv)
    # the gradient views share the sequence / batch structure of their forward partners; only the row stride may differ
    def strides(t, ref, ss, bs):
        return ss // ref.stride(0) * t.stride(0), bs // ref.stride(0) * t.stride(0)
    d.do_ss, d.do_bs = strides(dout, out, fwd.o_ss, fwd.o_bs)
    d.dq_ss, d.dq_bs = strides(dq, q, fwd.q_ss, fwd.q_bs)
    d.dk_ss, d.dk_bs = strides(dk, k, fwd.k_ss, fwd.k_bs)
    d.dv_ss, d.dv_bs = strides(dv, v, fwd.v_ss, fwd.v_bs)
    delta = torch.empty_like(lse)
    d.delta = _p(delta)
    check(_lib.load().seer_attn_bwd(C.byref(d), _stream()), "seer_attn_bwd")


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None,
            colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[N, K] (fp32) = a[M, N]^T @ b[M, K]: the weight gradient dY^T X with both operands in their token-major layout
    (row-strided views allowed); colsum[N] (optional, fp32) receives the column sums of a = the bias gradient."""
    _req(a, bf16, "a"); _req(b, bf16, "b")
    assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.stride(1) == 1 and b.stride(1) == 1
    M, N = a.shape
    K = b.shape[1]
    if out is None:
        out = torch.empty((N, K), device=a.device, dtype=torch.float32)
    _req(out, torch.float32, "out")
    assert out.shape == (N, K) and out.is_contiguous()
    lib = _lib.load()
    nbytes = lib.seer_gemm_tn_workspace_bytes(M, N, K)
    if nbytes < 0:
        check(int(nbytes), "seer_gemm_tn_workspace_bytes")
    ws = torch.empty((nbytes // 4,), device=a.device, dtype=torch.float32) if nbytes else None
    if colsum is not None:
        _req(colsum, torch.float32, "colsum")
        assert colsum.is_contiguous() and colsum.numel() == N
    check(lib.seer_gemm_tn_f32(_p(a), a.stride(0), _p(b), b.stride(0), M, N, K, _p(out), _p(colsum), _p(ws), nbytes, _stream()),
          "seer_gemm_tn_f32")
    return out


def gemm_tn_grouped(problems) -> None:
    """problems: [(a [M, N], b [M, K], out [N, K] fp32, colsum [N] fp32 or None), ...] -- out_i = a_i^T @ b_i (and colsum_i = the column
    sums of a_i) for all of them in ONE launch (+ one for the K slices of the long ones): the deferred weight gradients of a backward
    pass (seer_gemm_tn_grouped_f32).  A problem's result does not depend on what it is grouped with."""
    if not problems:
        return
    lib = _lib.load()
    items = (_lib.TnItem * len(problems))()
    for it, (a, b, out, colsum) in zip(items, problems):
        _req(a, bf16, "a"); _req(b, bf16, "b"); _req(out, torch.float32, "out")
        assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.stride(1) == 1 and b.stride(1) == 1
        M, N = a.shape
        K = b.shape[1]
        assert out.shape == (N, K) and out.is_contiguous()
        if colsum is not None:
            _req(colsum, torch.float32, "colsum")
            assert colsum.is_contiguous() and colsum.numel() == N
        it.A, it.B, it.C, it.colsum = _p(a), _p(b), _p(out), _p(colsum)
        it.lda, it.ldb, it.M, it.N, it.K = a.stride(0), b.stride(0), M, N, K
    nbytes = lib.seer_gemm_tn_grouped_workspace_bytes(items, len(problems))
    if nbytes < 0:
        check(int(nbytes), "seer_gemm_tn_grouped_workspace_bytes")
    ws = torch.empty((nbytes // 4,), device=problems[0][0].device, dtype=torch.float32) if nbytes else None
    check(lib.seer_gemm_tn_grouped_f32(items, len(problems), _p(ws), nbytes, _stream()), "seer_gemm_tn_grouped_f32")


def transpose(x: torch.Tensor, pad_to: int = 64) -> torch.Tensor:
    """x [rows, cols] (row-strided view) -> [cols, round_up(rows, pad_to)] with the pad columns zero."""
    _req(x, bf16, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    ldy = (rows + pad_to - 1) // pad_to * pad_to
    y = torch.empty((cols, ldy), device=x.device, dtype=bf16)
    check(_lib.load().seer_transpose_bf16(_p(x), rows, cols, x.stride(0), _p(y), ldy, _stream()), "seer_transpose_bf16")
    return y


class TransposePlan:
    """`run()` transposes every x_i [rows, cols] (bf16, row-strided views) into y_i [cols, round_up(rows, 64)] in ONE launch
    (seer_transpose_batched_bf16).  The outputs are views of one arena and live as long as the plan; the item table is built
    once, on the device, from the tensors' addresses -- the inputs must stay where they are (the trainer's flat bf16 weights do)."""

    def __init__(self, xs: "Sequence[torch.Tensor]"):
        assert len(xs) > 0
        dev = xs[0].device
        sizes, total = [], 0
        for x in xs:
            _req(x, bf16, "x")
            assert x.dim() == 2 and x.stride(1) == 1 and x.device == dev
            rows, cols = x.shape
            ldy = (rows + 63) // 64 * 64
            assert cols % 8 == 0 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0, "seer_transpose_batched_bf16: 16-byte rows"
            sizes.append((rows, cols, ldy, total))
            total += cols * ldy
        self.arena = torch.zeros((total,), device=dev, dtype=bf16)
        self.inputs = list(xs)                                     # keeps the sources alive: the table holds their addresses
        self.outputs, table, tile0 = [], [], 0
        for x, (rows, cols, ldy, off) in zip(xs, sizes):
            y = self.arena[off:off + cols * ldy].view(cols, ldy)
            assert y.data_ptr() % 16 == 0
            self.outputs.append(y)
            tiles_r = ldy // 64
            # seer_transpose_item: x, y, rows, ldy, tile0 (int64 x 5), cols, ldx, tiles_r, reserved (int32 x 4) = 56 bytes
            table.append((x.data_ptr(), y.data_ptr(), rows, ldy, tile0, cols | (x.stride(0) << 32), tiles_r))
            tile0 += tiles_r * ((cols + 63) // 64)
        self.total_tiles, self.n = tile0, len(table)
        self.table = torch.tensor(table, dtype=torch.int64).to(dev)   # [n, 7] int64 = the C struct (little endian)

    def run(self) -> None:
        check(_lib.load().seer_transpose_batched_bf16(_p(self.table), self.n, self.total_tiles, _stream()),
              "seer_transpose_batched_bf16")


def _col_ws(rows: int, cols: int, device) -> torch.Tensor:
    n = _lib.load().seer_colsum_workspace_floats(rows, cols)
    if n < 0:
        check(int(n), "seer_colsum_workspace_floats")
    return torch.empty((n,), device=device, dtype=torch.float32)


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, bf16, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols,), device=x.device, dtype=torch.float32)
    ws = _col_ws(rows, cols, x.device)
    check(_lib.load().seer_colsum_bf16(_p(x), rows, cols, x.stride(0), _p(out), _p(ws), _stream()), "seer_colsum_bf16")
    return out


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, *, eps: float = 1e-5,
                  dres: Optional[torch.Tensor] = None, dx: Optional[torch.Tensor] = None,
                  dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                  defer: Optional[list] = None) -> torch.Tensor:
    """defer = a list: d gamma / d beta are left as partial slabs and (slabs, n, d beta, d gamma) is appended to it -- colfinal_grouped
    adds the slabs of every norm on the list in one launch (same bits)."""
    _req(x, bf16, "x"); _req(dy, bf16, "dy"); _req(gamma, torch.float32, "gamma")
    assert x.dim() == 2 and x.stride(1) == 1 and dy.shape == x.shape and dy.stride(1) == 1
    rows, Cc = x.shape
    if dx is None:
        dx = torch.empty((rows, Cc), device=x.device, dtype=bf16)
    ws = _col_ws(rows, Cc, x.device) if dgamma is not None else None
    if dgamma is not None and defer is not None:
        _req(dgamma, torch.float32, "dgamma"); _req(dbeta, torch.float32, "dbeta")
        assert dgamma.is_contiguous() and dbeta.is_contiguous() and dgamma.numel() == Cc and dbeta.numel() == Cc
        lib = _lib.load()
        check(lib.seer_layernorm_bwd_partials(_p(x), _p(dy), rows, Cc, x.stride(0), dy.stride(0), _p(gamma), float(eps),
                                              _p(dres), 0 if dres is None else dres.stride(0), _p(dx), dx.stride(0), _p(ws), _stream()),
              "seer_layernorm_bwd_partials")
        defer.append((ws, int(lib.seer_layernorm_bwd_slabs(rows)), 2, Cc, dbeta, dgamma))
        return dx
    check(_lib.load().seer_layernorm_bwd(_p(x), _p(dy), rows, Cc, x.stride(0), dy.stride(0), _p(gamma), float(eps),
                                         _p(dres), 0 if dres is None else dres.stride(0), _p(dx), dx.stride(0),
                                         _p(dgamma), _p(dbeta), _p(ws), _stream()), "seer_layernorm_bwd")
    return dx


def colfinal_grouped(items) -> None:
    """items: [(slabs [n][NV][C] fp32, n, NV, C, out0 [C] or None, out1 [C] or None), ...]: out_v = the sum of the slabs, every item in
    one launch (seer_colfinal_grouped)"""
    if not items:
        return
    arr = (_lib.ColfinalItem * len(items))()
    for it, (ws, n, NV, Cc, o0, o1) in zip(arr, items):
        _req(ws, torch.float32, "slabs")
        assert ws.numel() >= n * NV * Cc
        it.ws, it.out0, it.out1, it.nblocks, it.NV, it.C = _p(ws), _p(o0), _p(o1), n, NV, Cc
    check(_lib.load().seer_colfinal_grouped(arr, len(items), _stream()), "seer_colfinal_grouped")


def groupnorm_bwd(x1: torch.Tensor, x2: Optional[torch.Tensor], batch: int, groups: int, stats: torch.Tensor, count: float,
                  eps: float, gamma: torch.Tensor, beta: torch.Tensor, silu: bool, dy: torch.Tensor, *,
                  dres1: Optional[torch.Tensor] = None, dres2: Optional[torch.Tensor] = None,
                  dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    _req(x1, bf16, "x1"); _req(dy, bf16, "dy")
    assert x1.is_contiguous() and dy.is_contiguous() and (x2 is None or x2.is_contiguous())
    rows = x1.shape[0] // batch
    C1 = x1.shape[1]
    C2 = 0 if x2 is None else x2.shape[1]
    assert dy.shape == (x1.shape[0], C1 + C2)
    for t in (dres1, dres2):
        assert t is None or t.is_contiguous()
    lib = _lib.load()
    n = lib.seer_groupnorm_bwd_workspace_floats(C1 + C2, batch, rows, groups)
    if n < 0:
        check(int(n), "seer_groupnorm_bwd_workspace_floats")
    ws = torch.empty((n,), device=x1.device, dtype=torch.float32)
    dx1 = torch.empty_like(x1)
    dx2 = None if x2 is None else torch.empty_like(x2)
    check(lib.seer_groupnorm_bwd(_p(x1), C1, _p(x2), C2, batch, rows, groups, _p(stats), float(count), float(eps), _p(gamma),
                                 _p(beta), int(silu), _p(dy), _p(dres1), _p(dres2), _p(dx1), _p(dx2), _p(dgamma), _p(dbeta),
                                 _p(ws), _stream()), "seer_groupnorm_bwd")
    return dx1, dx2


def geglu_fwd(pre: torch.Tensor) -> torch.Tensor:
    _req(pre, bf16, "pre")
    assert pre.dim() == 2 and pre.stride(1) == 1
    rows, two_inner = pre.shape
    out = torch.empty((rows, two_inner // 2), device=pre.device, dtype=bf16)
    check(_lib.load().seer_geglu_fwd(_p(pre), rows, two_inner // 2, pre.stride(0), _p(out), out.stride(0), _stream()),
          "seer_geglu_fwd")
    return out


def geglu_bwd(pre: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    _req(pre, bf16, "pre"); _req(dout, bf16, "dout")
    rows, two_inner = pre.shape
    assert dout.shape == (rows, two_inner // 2) and dout.stride(1) == 1 and pre.stride(1) == 1
    dpre = torch.empty((rows, two_inner), device=pre.device, dtype=bf16)
    check(_lib.load().seer_geglu_bwd(_p(pre), _p(dout), rows, two_inner // 2, pre.stride(0), dout.stride(0), _p(dpre),
                                     dpre.stride(0), _stream()), "seer_geglu_bwd")
    return dpre


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(a, bf16, "a"); _req(b, bf16, "b")
    assert a.dim() == 2 and a.shape == b.shape and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty(a.shape, device=a.device, dtype=bf16)
    check(_lib.load().seer_add_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), a.shape[0], a.shape[1],
                                    _stream()), "seer_add_bf16")
    return out


def sumpool2x(du: torch.Tensor, n_img: int, H: int, W: int) -> torch.Tensor:
    """du [n_img*2H*2W, C] -> [n_img*H*W, C]"""
    _req(du, bf16, "du")
    assert du.is_contiguous() and du.shape[0] == n_img * 4 * H * W
    dx = torch.empty((n_img * H * W, du.shape[1]), device=du.device, dtype=bf16)
    check(_lib.load().seer_sumpool2x_bf16(_p(du), n_img, H, W, du.shape[1], _p(dx), _stream()), "seer_sumpool2x_bf16")
    return dx


def zero_insert2x(d: torch.Tensor, n_img: int, H: int, W: int) -> torch.Tensor:
    """d [n_img*H*W, C] -> [n_img*2H*2W, C]"""
    _req(d, bf16, "d")
    assert d.is_contiguous() and d.shape[0] == n_img * H * W
    z = torch.empty((n_img * 4 * H * W, d.shape[1]), device=d.device, dtype=bf16)
    check(_lib.load().seer_zero_insert2x_bf16(_p(d), n_img, H, W, d.shape[1], _p(z), _stream()), "seer_zero_insert2x_bf16")
    return z


def mse_loss_grad(pred: torch.Tensor, target: torch.Tensor, cond_f: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pred [B, C, F_total, H, W] fp32, target [B, C, F_total - cond_f, H, W] fp32 -> (loss [1], dpred like pred)."""
    _req(pred, torch.float32, "pred"); _req(target, torch.float32, "target")
    assert pred.is_contiguous() and target.is_contiguous()
    B, Cc, Ft, H, W = pred.shape
    assert target.shape == (B, Cc, Ft - cond_f, H, W)
    loss = torch.empty((1,), device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    ws = torch.empty((1024,), device=pred.device, dtype=torch.float32)
    check(_lib.load().seer_mse_loss_grad(_p(pred), _p(target), B, Cc, Ft, cond_f, H * W, _p(loss), _p(dpred), _p(ws),
                                         _stream()), "seer_mse_loss_grad")
    return loss, dpred


def text_loss_grad(y: torch.Tensor, target: torch.Tensor, b: int, Fr: int, dy: torch.Tensor) -> torch.Tensor:
    """y, dy bf16 [b*F*l, C] (rows (b, f, l)); target fp32 [b, l, C].  Adds the text-loss gradient to dy, returns the loss [1]."""
    _req(y, bf16, "y"); _req(dy, bf16, "dy"); _req(target, torch.float32, "target")
    assert y.is_contiguous() and dy.is_contiguous() and target.is_contiguous() and y.shape == dy.shape
    LC = target.numel() // b
    assert y.numel() == b * Fr * LC
    loss = torch.empty((1,), device=y.device, dtype=torch.float32)
    ws = torch.empty((1024,), device=y.device, dtype=torch.float32)
    check(_lib.load().seer_text_loss_grad(_p(y), _p(target), b, Fr, LC, _p(dy), _p(loss), _p(ws), _stream()), "seer_text_loss_grad")
    return loss


def conv_out_bwd(dpred: torch.Tensor, w_ohwc: torch.Tensor) -> torch.Tensor:
    """dpred [B, Cout, F, H, W] fp32, w fp32 [Cout, 3, 3, C0] -> dx bf16 [B*F*H*W, C0]"""
    _req(dpred, torch.float32, "dpred"); _req(w_ohwc, torch.float32, "w")
    assert dpred.is_contiguous() and w_ohwc.is_contiguous()
    B, Cout, F, H, W = dpred.shape
    C0 = w_ohwc.shape[-1]
    dx = torch.empty((B * F * H * W, C0), device=dpred.device, dtype=bf16)
    check(_lib.load().seer_conv_out_bwd(_p(dpred), B, C0, F, H, W, _p(w_ohwc), Cout, _p(dx), _stream()), "seer_conv_out_bwd")
    return dx


def axpby(y: torch.Tensor, x: torch.Tensor, alpha: float, beta: float) -> None:
    """y = beta*y + alpha*x (flat fp32 buffers)"""
    _req(y, torch.float32, "y"); _req(x, torch.float32, "x")
    assert y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel()
    check(_lib.load().seer_axpby_f32(_p(y), _p(x), float(alpha), float(beta), y.numel(), _stream()), "seer_axpby_f32")


def sumsq(g: torch.Tensor) -> torch.Tensor:
    _req(g, torch.float32, "g")
    assert g.is_contiguous()
    out = torch.empty((1,), device=g.device, dtype=torch.float32)
    ws = torch.empty((1024,), device=g.device, dtype=torch.float32)
    check(_lib.load().seer_sumsq_f32(_p(g), g.numel(), _p(out), _p(ws), _stream()), "seer_sumsq_f32")
    return out


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, lr: float, betas=(0.9, 0.999),
               eps: float = 1e-8, weight_decay: float = 1e-2, step: int, grad_sumsq: Optional[torch.Tensor] = None,
               max_norm: float = 1.0, p_bf16: Optional[torch.Tensor] = None) -> None:
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, torch.float32, n)
        assert t.is_contiguous() and t.numel() == p.numel()
    if p_bf16 is not None:
        _req(p_bf16, bf16, "p_bf16")
        assert p_bf16.is_contiguous() and p_bf16.numel() == p.numel()
    check(_lib.load().seer_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                                      float(eps), float(weight_decay), int(step), _p(grad_sumsq), float(max_norm), _p(p_bf16),
                                      _stream()), "seer_adamw_step")
