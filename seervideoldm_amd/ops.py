"""Thin tensor-level wrappers over the C ABI (include/seer_hip.h).

torch supplies device memory and the current stream; every FLOP below runs in libseer_hip.so.  Tensors must live
on a ROCm device -- there is no CPU path (a CPU tensor raises).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import AttnDesc, GemmDesc, check

bf16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.SeerHipError("seervideoldm_amd kernels need tensors on a ROCm device (no CPU fallback)")
    return t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise _lib.SeerHipError(f"{name}: tensor must be on a ROCm device (no CPU fallback)")


f16 = torch.float16


def _req16(t: torch.Tensor, name: str, like: Optional[torch.Tensor] = None) -> int:
    """16-bit activations / weights: bf16 (the UNet path) or IEEE half (the VAE path, SEER_EPI_F16 / SEER_DT_F16); all 16-bit
    operands of one launch share the type.  Returns the SEER_DT_* code."""
    if t.dtype not in (bf16, f16):
        raise TypeError(f"{name}: expected torch.bfloat16 or torch.float16, got {t.dtype}")
    if like is not None and t.dtype != like.dtype:
        raise TypeError(f"{name}: {t.dtype} next to {like.dtype}: the 16-bit operands of a launch share one type")
    if not t.is_cuda:
        raise _lib.SeerHipError(f"{name}: tensor must be on a ROCm device (no CPU fallback)")
    return _lib.SEER_DT_F16 if t.dtype == f16 else _lib.SEER_DT_BF16


class ColSums:
    """Per-tile column sums a GEMM / conv left next to its output (seer_gemm_desc::colsum): buf [phases, tiles, C, 2] fp32 =
    (sum, sum of squares) of the stored bf16 values over each tile's rows.  groupnorm_stats_from_colsums turns them into
    GroupNorm statistics without a pass over the activations."""
    __slots__ = ("buf", "C", "phases", "tiles")

    def __init__(self, buf: torch.Tensor, C_: int, phases: int, tiles: int):
        self.buf, self.C, self.phases, self.tiles = buf, C_, phases, tiles


class ColSumsFx:
    """Column sums ACCUMULATED per batch element in 64-bit fixed point (seer_gemm_desc::colsum_fx): buf [reps, batch, 2, C] int64
    out of an FxArena (planes: sum, sum of squares at scale 2^20; the replicas are added by the reader).  groupnorm_apply_fx
    normalises with them in one launch."""
    __slots__ = ("buf", "C", "reduced")

    def __init__(self, buf: torch.Tensor, C_: int):
        self.buf, self.C = buf, C_
        self.reduced = False        # frame shards: the sums of all shards have been added in (parallel.FrameShard.reduce_fx)

    @property
    def reps(self) -> int:
        return self.buf.shape[0]

    def totals(self) -> torch.Tensor:
        """[batch, C, 2] fp64 (sum, sum of squares)"""
        return self.buf.sum(dim=0).permute(0, 2, 1).to(torch.float64) / float(1 << 20)


class RowStats:
    """Per-row (sum, sum of squares) a GEMM accumulated next to its output (seer_gemm_desc::rowstat): buf [M, 2] int64 at scale
    2^24, out of an FxArena (zeroed) or a fresh zeroed tensor.  gemm(..., ln=(rowstats, wsum, eps)) normalises the rows inside
    the consuming GEMM (folded LayerNorm)."""
    __slots__ = ("buf",)

    def __init__(self, buf: torch.Tensor):
        self.buf = buf

    def totals(self) -> torch.Tensor:
        return self.buf.to(torch.float64) / float(1 << 24)


def fold_layernorm(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor] = None, dtype=bf16):
    """LayerNorm(gamma, beta) followed by Linear(w [N, K], bias) as ONE GEMM over the un-normalised rows:
    LN(x) W^T + b = rstd * (x W'^T - mean * wsum) + b',  W' = bf16(gamma (.) W),  wsum = row sums of W' AS ROUNDED (the mean
    cancels exactly),  b' = W beta + b (fp32).  Returns (W' bf16 [N, K], wsum fp32 [N], b' fp32 [N]); `dtype` = the engine's 16-bit
    storage type (torch.float16 under fp16 autocast)."""
    wf = w.float()
    wp = (wf * gamma.float()[None, :]).to(dtype).contiguous()
    wsum = wp.float().sum(dim=1).contiguous()
    bp = wf @ beta.float()
    if bias is not None:
        bp = bp + bias.float()
    return wp, wsum, bp.contiguous()


class FxArena:
    """Bump allocator over one int64 buffer for the ColSumsFx of a UNet evaluation: reset() once per evaluation zeroes everything
    that has EVER been handed out (one fill launch), take() hands out [batch, C, 2] slices.  Pass colsum_batch=(B, arena) to
    gemm / conv3x3 / conv_up2x.
    The zeroed prefix never shrinks: a replayed hipGraph adds into the slots baked into it without this object seeing it, so the
    slots a larger shape took once may hold sums again whatever the evaluations in between asked for (capture shape A, run a
    smaller shape eagerly, replay A, run A eagerly: that last reset must cover A's slots, not the small shape's)."""

    def __init__(self, device, int64_elems: int):
        self.buf = torch.zeros((int64_elems,), device=device, dtype=torch.int64)
        self.used = 0
        self.high = 0                  # high-water mark over the arena's whole life

    def reset(self):
        if self.high:
            self.buf[:self.high].zero_()
        self.used = 0

    def take_rows(self, rows: int) -> Optional[torch.Tensor]:
        n = rows * 2
        if self.used + n > self.buf.numel():
            return None
        t = self.buf[self.used:self.used + n].view(rows, 2)
        self.used += n
        self.high = max(self.high, self.used)
        return t

    def take(self, reps: int, batch: int, C_: int) -> Optional[torch.Tensor]:
        n = reps * batch * C_ * 2
        if self.used + n > self.buf.numel():
            return None
        t = self.buf[self.used:self.used + n].view(reps, batch, 2, C_)
        self.used += n
        self.high = max(self.high, self.used)
        return t


# ------------------------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, w: torch.Tensor, *, bias=None, residual=None, rowvec=None, rows_per_batch=0,
         a2: Optional[torch.Tensor] = None, geglu=False, silu=False, out_f32=False, out: Optional[torch.Tensor] = None,
         tile=0, splits=0, rotary=None, col_scale=None, colsum_batch=0, rowstat=False, ln=None) -> Optional[torch.Tensor]:
    """out[M,N] = epi(a[M,K1] | a2[M,K-K1]) @ w[N,K]^T ; a/a2 may be row-strided views (last dim contiguous).
    rowstat = True or an FxArena: out.rowstats = RowStats of the output rows (None when this launch cannot accumulate them).
    ln = (RowStats of a, wsum, eps): a holds UN-normalised rows and w / bias are fold_layernorm()'s W' / b' -- the LayerNorm is
    applied inside the epilogue.  Returns None (nothing launched) when this launch cannot fold: run layernorm() + the plain weights.
    colsum_batch = B > 0: the output feeds a GroupNorm over B batch elements -- out.colsums is set to the ColSums the launch
    left (or None when this launch cannot produce them; the caller then runs groupnorm_stats on the output).
    rotary = (cos_sin table, tokens_per_batch, pos_offset, head_dim, rot_dim, cols): rotate columns < cols in the epilogue.
    col_scale = (factor, cols): multiply output columns < cols by factor (the q columns of a projection carry the softmax
    scale * log2(e) for attention(..., q_prescaled=True))."""
    dt = _req16(a, "a"); _req16(w, "w", a)
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.is_contiguous()
    M, K1 = a.shape
    N, K = w.shape
    d = GemmDesc()
    d.A, d.W = _p(a), _p(w)
    d.lda = a.stride(0)
    if a2 is not None:
        _req16(a2, "a2", a)
        assert a2.shape[0] == M and a2.stride(1) == 1 and K1 + a2.shape[1] == K
        d.A2, d.lda2 = _p(a2), a2.stride(0)
    else:
        assert K1 == K, f"K mismatch {K1} vs {K}"
    d.M, d.N, d.K, d.K1 = M, N, K, K1
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    assert out.shape == (M, n_out) and out.stride(1) == 1
    d.C, d.ldc = _p(out), out.stride(0)
    if bias is not None:
        _req(bias, torch.float32, "bias"); d.bias = _p(bias)
    if residual is not None:
        _req16(residual, "residual", a)
        assert residual.shape == (M, n_out) and residual.stride(1) == 1
        d.residual, d.ldr = _p(residual), residual.stride(0)
    if rowvec is not None:
        _req(rowvec, torch.float32, "rowvec")
        assert rowvec.dim() == 2 and rowvec.stride(1) == 1
        d.rowvec, d.rowvec_ld, d.rows_per_batch = _p(rowvec), rowvec.stride(0), rows_per_batch
    d.mode = _lib.SEER_GEMM_PLAIN
    d.epilogue = (_lib.SEER_EPI_GEGLU if geglu else 0) | (_lib.SEER_EPI_SILU if silu else 0) | \
                 (_lib.SEER_EPI_OUT_F32 if (out.dtype == torch.float32) else 0) | (_lib.SEER_EPI_F16 if dt else 0)
    assert out.dtype in (torch.float32, a.dtype)
    if rotary is not None:
        table, tpb, pos_off, hd, rd, cols = rotary
        _req(table, torch.float32, "rotary table")
        assert table.shape[0] >= tpb + pos_off and table.shape[1] * 2 == rd
        d.epilogue |= _lib.SEER_EPI_ROTARY
        d.rot_table, d.rot_tokens_per_batch, d.rot_pos_offset = _p(table), tpb, pos_off
        d.rot_head_dim, d.rot_dim, d.rot_cols = hd, rd, cols
    if col_scale is not None:
        d.epilogue |= _lib.SEER_EPI_COLSCALE
        d.col_scale, d.col_scale_cols = float(col_scale[0]), int(col_scale[1])
    d.batch = 1
    d.tile = tile
    d.splits = splits
    lib = _lib.load()
    if ln is not None:
        rs, wsum, eps = ln
        _req(wsum, torch.float32, "ln wsum")
        assert rs.buf.shape[0] == M and wsum.shape[0] == N and a2 is None
        d.ln_rowstat, d.ln_wsum, d.ln_eps = _p(rs.buf), _p(wsum), float(eps)
        if not lib.seer_gemm_lnfold_ok(C.byref(d)):
            return None
    rstats = None
    if rowstat and lib.seer_gemm_rowstat_ok(C.byref(d)):
        buf = rowstat.take_rows(M) if isinstance(rowstat, FxArena) else None      # rowstat = True or the evaluation's arena
        if buf is None:
            buf = torch.zeros((M, 2), device=a.device, dtype=torch.int64)
        rstats = RowStats(buf)
        d.rowstat = _p(buf)
    cs = _launch_gemm(d, a.device, "seer_gemm_bf16", colsum_batch)
    out.colsums = cs            # always assigned: a reused `out=` tensor must not keep the column sums of an earlier launch
    out.rowstats = rstats       # likewise
    return out


_SYNC_BYTES = 1 << 16          # counters of the in-launch split-K reduction: two 32-bit words per output tile
_sync_buffers: dict = {}


def _sync_buffer(device) -> torch.Tensor:
    """seer_gemm_desc::sync: zeroed ONCE per device; every launch leaves it zero again, and the engines launch on one stream
    (launches that could overlap in time would need buffers of their own).  Created outside any graph capture: the first
    call of a shape is always an eager warm-up, and a memset inside a captured step would replay with every step."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    buf = _sync_buffers.get(key)
    if buf is None:
        assert not torch.cuda.is_current_stream_capturing(), "the split-K counter buffer must exist before a graph capture"
        buf = torch.zeros((_SYNC_BYTES,), device=device, dtype=torch.uint8)
        _sync_buffers[key] = buf
    return buf


def _launch_gemm(d: GemmDesc, device, what: str, colsum_batch: int = 0) -> Optional[ColSums]:
    """split-K needs a caller-provided fp32 workspace (the library never allocates): ask, allocate, launch.
    colsum_batch > 0: also ask for per-tile column sums when the launch can produce them and no tile straddles two of the
    colsum_batch batch elements the M rows (of each phase) are split into."""
    lib = _lib.load()
    nbytes = lib.seer_gemm_workspace_bytes(C.byref(d))
    if nbytes < 0:
        check(int(nbytes), what)
    if nbytes > 0:
        ws = torch.empty((nbytes // 4,), device=device, dtype=torch.float32)
        d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    nsync = lib.seer_gemm_sync_bytes(C.byref(d))
    if nsync < 0:
        check(int(nsync), what)
    if 0 < nsync <= _SYNC_BYTES:
        d.sync, d.sync_bytes = _sync_buffer(device).data_ptr(), _SYNC_BYTES
    cs = None
    arena = None
    if isinstance(colsum_batch, tuple):
        colsum_batch, arena = colsum_batch
    if colsum_batch > 0 and d.M % colsum_batch == 0:
        fx = None
        if arena is not None:
            reps = C.c_int32(1)
            if lib.seer_gemm_colsum_fx_layout(C.byref(d), d.M // colsum_batch, C.byref(reps)) > 0:
                fx = arena.take(reps.value, colsum_batch, int(d.N))
        if fx is not None:
            d.colsum_fx, d.colsum_fx_rows, d.colsum_fx_reps = fx.data_ptr(), d.M // colsum_batch, fx.shape[0]
            cs = ColSumsFx(fx, int(d.N))
        rows = 0 if fx is not None else lib.seer_gemm_colsum_rows(C.byref(d))
        if rows > 0 and (d.M // colsum_batch) % rows == 0:
            phases, tiles = max(int(d.batch), 1), d.M // rows
            buf = torch.empty((phases, tiles, d.N, 2), device=device, dtype=torch.float32)
            d.colsum = buf.data_ptr()
            cs = ColSums(buf, int(d.N), phases, tiles)
    check(lib.seer_gemm_bf16(C.byref(d), _stream()), what)
    return cs


def gemm_batched(a: torch.Tensor, w: torch.Tensor, *, trans_out=False, out: Optional[torch.Tensor] = None,
                 bias: Optional[torch.Tensor] = None, out_f32=False, tile=0, col_scale=None) -> torch.Tensor:
    """a [Bt, M, K], w [Bt, N, K] (or [N, K] shared) -> out [Bt, M, N] (or [Bt, N, M] with trans_out).
    col_scale = (factor, cols) as in gemm()."""
    dt = _req16(a, "a"); _req16(w, "w", a)
    assert a.dim() == 3 and a.is_contiguous() and w.is_contiguous()
    Bt, M, K = a.shape
    N = w.shape[-2]
    d = GemmDesc()
    d.A, d.W = _p(a), _p(w)
    d.lda = K
    d.M, d.N, d.K, d.K1 = M, N, K, K
    if out is None:
        out = torch.empty((Bt, N, M) if trans_out else (Bt, M, N), device=a.device,
                          dtype=torch.float32 if out_f32 else a.dtype)
    d.C = _p(out)
    if bias is not None:
        _req(bias, torch.float32, "bias"); d.bias = _p(bias)
    d.ldc = M if trans_out else N
    d.batch = Bt
    d.strideA = M * K
    d.strideW = N * K if w.dim() == 3 else 0
    d.strideC = M * N
    d.epilogue = (_lib.SEER_EPI_TRANS_OUT if trans_out else 0) | (_lib.SEER_EPI_OUT_F32 if out.dtype == torch.float32 else 0) | \
                 (_lib.SEER_EPI_F16 if dt else 0)
    if col_scale is not None:
        d.epilogue |= _lib.SEER_EPI_COLSCALE
        d.col_scale, d.col_scale_cols = float(col_scale[0]), int(col_scale[1])
    d.tile = tile
    check(_lib.load().seer_gemm_bf16(C.byref(d), _stream()), "seer_gemm_bf16(batched)")
    return out


def conv3x3(x: torch.Tensor, w: torch.Tensor, n_img: int, Hin: int, Win: int, *, stride=1, upsample=False,
            bias=None, residual=None, rowvec=None, rows_per_batch=0, out: Optional[torch.Tensor] = None,
            tile=0, splits=0, pad_after_only=False, colsum_batch=0) -> torch.Tensor:
    """x: channels-last [n_img*Hin*Win, Cin] bf16; w: [Cout, 9*Cin] ((ky,kx,ci) order). Returns [n_img*Ho*Wo, Cout].
    pad_after_only: zero padding of one row / column only after the image (the VAE encoder's Downsample)."""
    dt = _req16(x, "x"); _req16(w, "w", x)
    assert x.is_contiguous() and w.is_contiguous()
    Cin = x.shape[1]
    Cout, K = w.shape
    assert K == 9 * Cin and x.shape[0] == n_img * Hin * Win
    Hs, Ws = (2 * Hin, 2 * Win) if upsample else (Hin, Win)
    pad = 1 if pad_after_only else 2
    Ho, Wo = (Hs + pad - 3) // stride + 1, (Ws + pad - 3) // stride + 1
    M = n_img * Ho * Wo
    d = GemmDesc()
    d.A, d.W = _p(x), _p(w)
    d.M, d.N, d.K, d.K1 = M, Cout, K, K
    if out is None:
        out = torch.empty((M, Cout), device=x.device, dtype=x.dtype)
    d.C, d.ldc = _p(out), out.stride(0)
    if bias is not None:
        _req(bias, torch.float32, "bias"); d.bias = _p(bias)
    if residual is not None:
        _req16(residual, "residual", x)
        assert residual.shape == (M, Cout) and residual.stride(1) == 1
        d.residual, d.ldr = _p(residual), residual.stride(0)
    if rowvec is not None:
        _req(rowvec, torch.float32, "rowvec")
        d.rowvec, d.rowvec_ld, d.rows_per_batch = _p(rowvec), rowvec.stride(0), rows_per_batch
    d.mode = _lib.SEER_GEMM_CONV3X3
    d.epilogue = _lib.SEER_EPI_F16 if dt else 0
    d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride, d.upsample = Hin, Win, Cin, Ho, Wo, stride, int(upsample)
    d.pad_after_only = int(pad_after_only)
    d.batch = 1
    d.tile = tile
    d.splits = splits
    cs = _launch_gemm(d, x.device, "seer_gemm_bf16(conv3x3)", colsum_batch)
    out.colsums = cs            # always assigned: a reused `out=` tensor must not keep the column sums of an earlier launch
    return out


def conv_up2x(x: torch.Tensor, w4: torch.Tensor, n_img: int, Hin: int, Win: int, *, bias=None,
              out: Optional[torch.Tensor] = None, tile=0, colsum_batch=0) -> torch.Tensor:
    """nearest-2x upsample + conv3x3 (Upsample3D, resnet.py:52-57) as four 2x2 phase convs in one launch.
    x: channels-last [n_img*Hin*Win, Cin] bf16; w4: [4, Cout, 4*Cin] from weights.pack_conv3x3_up_phases.
    Returns [n_img*2Hin*2Win, Cout]."""
    dt = _req16(x, "x"); _req16(w4, "w4", x)
    assert x.is_contiguous() and w4.is_contiguous()
    Cin = x.shape[1]
    four, Cout, K = w4.shape
    assert four == 4 and K == 4 * Cin and x.shape[0] == n_img * Hin * Win
    d = GemmDesc()
    d.A, d.W = _p(x), _p(w4)
    d.M, d.N, d.K, d.K1 = n_img * Hin * Win, Cout, K, K
    if out is None:
        out = torch.empty((n_img * 4 * Hin * Win, Cout), device=x.device, dtype=x.dtype)
    d.C, d.ldc = _p(out), out.stride(0)
    if bias is not None:
        _req(bias, torch.float32, "bias"); d.bias = _p(bias)
    d.mode = _lib.SEER_GEMM_CONV3X3
    d.epilogue = _lib.SEER_EPI_F16 if dt else 0
    d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride, d.upsample = Hin, Win, Cin, 2 * Hin, 2 * Win, 1, 2
    d.batch = 4
    d.tile = tile
    d.splits = 1
    cs = _launch_gemm(d, x.device, "seer_gemm_bf16(conv_up2x)", colsum_batch)
    out.colsums = cs            # always assigned: a reused `out=` tensor must not keep the column sums of an earlier launch
    return out


# ------------------------------------------------------------------------------------------------------------
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, batch: int, heads: int,
              head_dim: int, Sq: int, Sk: int, causal=False, scale: Optional[float] = None,
              window=None, Fq: Optional[int] = None, causal_offset: int = 0,
              seq_stride_rows: int = 1, batch_stride_rows: Optional[int] = None,
              lse: Optional[torch.Tensor] = None, q_prescaled: bool = False, variant: int = 0,
              q_head_major: bool = False, kv_head_major: bool = False, _desc_only: bool = False) -> torch.Tensor:
    """q/k/v/out are 2-D token-major views [batch*S, >=heads*head_dim] (row stride = token stride, e.g. column slices
    of a fused qkv buffer).  window = (ws, F, H, W) selects the temporal window form (K/V: F*H*W tokens per batch
    element in memory, Sk = F*ws*ws per window; Q/O hold Fq frames, Fq = F unless frame-sharded).  causal_offset is the
    sequence position of query 0 in the key sequence (frame shards).  seq_stride_rows / batch_stride_rows: token s of
    sequence b sits in row b*batch_stride_rows + s*seq_stride_rows (default: sequences stored one after the other) --
    FSTextTransformer's attention over frames reads rows ordered (frame, token) with seq stride = tokens per frame.
    q_head_major / kv_head_major: that operand is HEAD-MAJOR, [batch][head][tokens][head_dim] contiguous, passed as its 2-D view
    [batch * heads * tokens, head_dim] (seer_attn_desc::q_hs / k_hs / v_hs); `out` is always token-major.
    q_prescaled: q already holds q * scale * log2(e) (gemm(..., col_scale=(qk_prescale(head_dim), cols))).
    variant: kernel selection for A/B runs (include/seer_hip.h, seer_attn_desc.variant); 0 = auto."""
    dt = _req16(q, "q")
    for t, n in ((k, "k"), (v, "v"), (out, "out")):
        _req16(t, n, q)
    for t in (q, k, v, out):
        assert t.dim() == 2 and t.stride(1) == 1
    d = AttnDesc()
    d.Q, d.K, d.V, d.O = _p(q), _p(k), _p(v), _p(out)
    d.q_ss, d.k_ss, d.v_ss, d.o_ss = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    d.causal_offset = causal_offset
    if window is None:
        tq, tk = Sq, Sk
    else:
        ws, F, H, W = window
        Fq = F if Fq is None else Fq
        tq, tk = Fq * H * W, F * H * W
        d.window_ws, d.F, d.H, d.W, d.Fq = ws, F, H, W, Fq
    assert q.shape[0] == batch * tq * (heads if q_head_major else 1) and out.shape[0] == batch * tq
    assert k.shape[0] == v.shape[0] == batch * tk * (heads if kv_head_major else 1)
    d.q_bs, d.o_bs = tq * q.stride(0), tq * out.stride(0)
    d.k_bs, d.v_bs = tk * k.stride(0), tk * v.stride(0)
    if q_head_major:
        assert q.is_contiguous() and q.shape[1] == head_dim
        d.q_hs, d.q_bs = tq * head_dim, heads * tq * head_dim
    if kv_head_major:
        assert k.is_contiguous() and v.is_contiguous() and k.shape[1] == v.shape[1] == head_dim
        d.k_hs = d.v_hs = tk * head_dim
        d.k_bs = d.v_bs = heads * tk * head_dim
    if seq_stride_rows != 1 or batch_stride_rows is not None:
        assert window is None and batch_stride_rows is not None and not (q_head_major or kv_head_major)
        d.q_ss, d.k_ss, d.v_ss, d.o_ss = (seq_stride_rows * t.stride(0) for t in (q, k, v, out))
        d.q_bs, d.k_bs, d.v_bs, d.o_bs = (batch_stride_rows * t.stride(0) for t in (q, k, v, out))
    d.batch, d.heads, d.head_dim, d.Sq, d.Sk = batch, heads, head_dim, Sq, Sk
    d.causal = int(causal)
    d.scale = float(scale if scale is not None else head_dim ** -0.5)
    d.flags = (_lib.SEER_ATTN_Q_PRESCALED if q_prescaled else 0) | (_lib.SEER_ATTN_F16 if dt else 0)
    d.variant = variant
    if lse is not None:         # training: keep the softmax statistics for seer_attn_bwd
        _req(lse, torch.float32, "lse")
        nb = batch if window is None else batch * (window[2] // window[0]) * (window[3] // window[0])
        assert lse.is_contiguous() and lse.numel() == nb * heads * Sq
        d.lse = _p(lse)
    if _desc_only:
        return d
    check(_lib.load().seer_attn_fwd(C.byref(d), _stream()), "seer_attn_fwd")
    return out


LOG2E = 1.4426950408889634


def qk_prescale(head_dim: int, scale: Optional[float] = None) -> float:
    """the factor a projection's q columns are multiplied by for attention(..., q_prescaled=True)"""
    return (head_dim ** -0.5 if scale is None else scale) * LOG2E


def rotary_table(freqs: torch.Tensor, T: int) -> torch.Tensor:
    _req(freqs, torch.float32, "freqs")
    half = freqs.numel()
    cs = torch.empty((T, half, 2), device=freqs.device, dtype=torch.float32)
    check(_lib.load().seer_rotary_table(_p(freqs), T, half, _p(cs), _stream()), "seer_rotary_table")
    return cs


def rotary_inplace(x: torch.Tensor, col0_q: int, col0_k: int, heads: int, head_dim: int, rot_dim: int,
                   tokens_per_batch: int, cos_sin: torch.Tensor, pos_offset: int = 0) -> None:
    _req(x, bf16, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    assert cos_sin.shape[0] >= tokens_per_batch + pos_offset
    check(_lib.load().seer_rotary_inplace(_p(x), x.shape[0], x.stride(0), col0_q, col0_k, heads, head_dim, rot_dim,
                                          tokens_per_batch, pos_offset, _p(cos_sin), _stream()), "seer_rotary_inplace")


# ------------------------------------------------------------------------------------------------------------
def groupnorm_stats(x1: torch.Tensor, x2: Optional[torch.Tensor], batch: int, groups: int,
                    stats: torch.Tensor) -> torch.Tensor:
    """(sum, sumsq) per (b, g) -> stats [batch, groups, 2] fp32 (overwritten; deterministic two-stage reduction)."""
    dt = _req16(x1, "x1")
    if x2 is not None:
        _req16(x2, "x2", x1)
    assert x1.is_contiguous() and (x2 is None or x2.is_contiguous())
    rows = x1.shape[0] // batch
    C2 = 0 if x2 is None else x2.shape[1]
    lib = _lib.load()
    nws = lib.seer_groupnorm_workspace_floats(x1.shape[1] + C2, batch, rows, groups)
    if nws < 0:
        check(int(nws), "seer_groupnorm_workspace_floats")
    ws = torch.empty((nws,), device=x1.device, dtype=torch.float32)
    check(lib.seer_groupnorm_stats_dt(_p(x1), x1.shape[1], _p(x2), C2, batch, rows, groups, _p(stats), _p(ws), dt,
                                   _stream()), "seer_groupnorm_stats")
    return stats


def groupnorm_stats_from_colsums(cs1: ColSums, cs2: Optional[ColSums], batch: int, groups: int,
                                 stats: torch.Tensor) -> torch.Tensor:
    """The statistics of groupnorm_stats from the column sums the producers of x1 (and its concat partner x2) left."""
    check(_lib.load().seer_groupnorm_stats_from_colsums(
        _p(cs1.buf), cs1.C, cs1.phases, cs1.tiles, _p(cs2.buf) if cs2 is not None else None,
        cs2.C if cs2 is not None else 0, cs2.phases if cs2 is not None else 0, cs2.tiles if cs2 is not None else 0,
        batch, groups, _p(stats), _stream()), "seer_groupnorm_stats_from_colsums")
    return stats


def groupnorm_apply(x1: torch.Tensor, x2: Optional[torch.Tensor], batch: int, groups: int, stats: torch.Tensor,
                    count: float, eps: float, gamma: torch.Tensor, beta: torch.Tensor, silu: bool,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows = x1.shape[0] // batch
    C2 = 0 if x2 is None else x2.shape[1]
    Ct = x1.shape[1] + C2
    dt = _req16(x1, "x1")
    if out is None:
        out = torch.empty((x1.shape[0], Ct), device=x1.device, dtype=x1.dtype)
    _req16(out, "out", x1)
    _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    check(_lib.load().seer_groupnorm_apply_dt(_p(x1), x1.shape[1], _p(x2), C2, batch, rows, groups, _p(stats),
                                              float(count), float(eps), _p(gamma), _p(beta), int(silu), _p(out), dt,
                                              _stream()), "seer_groupnorm_apply")
    return out


def groupnorm_apply_from_colsums(x1: torch.Tensor, x2: Optional[torch.Tensor], cs1: ColSums, cs2: Optional[ColSums], batch: int,
                                 groups: int, count: float, eps: float, gamma: torch.Tensor, beta: torch.Tensor, silu: bool,
                                 out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """groupnorm_stats_from_colsums + groupnorm_apply as one launch (single-process engines).  Returns None when this channel
    layout has to stay on the two calls (SEER_ENOSYS)."""
    dt = _req16(x1, "x1")
    if x2 is not None:
        _req16(x2, "x2", x1)
    rows = x1.shape[0] // batch
    C2 = 0 if x2 is None else x2.shape[1]
    if out is None:
        out = torch.empty((x1.shape[0], x1.shape[1] + C2), device=x1.device, dtype=x1.dtype)
    _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    rc = _lib.load().seer_groupnorm_apply_from_colsums_dt(
        _p(x1), x1.shape[1], _p(x2), C2, _p(cs1.buf), cs1.phases, cs1.tiles, _p(cs2.buf) if cs2 is not None else None,
        cs2.phases if cs2 is not None else 0, cs2.tiles if cs2 is not None else 0, batch, rows, groups, float(count), float(eps),
        _p(gamma), _p(beta), int(silu), _p(out), dt, _stream())
    if rc == _lib.SEER_ENOSYS:
        return None
    check(rc, "seer_groupnorm_apply_from_colsums")
    return out


def groupnorm_apply_fx(x1: torch.Tensor, x2: Optional[torch.Tensor], fx1: ColSumsFx, fx2: Optional[ColSumsFx], batch: int,
                       groups: int, count: float, eps: float, gamma: torch.Tensor, beta: torch.Tensor, silu: bool,
                       out: Optional[torch.Tensor] = None, stats_out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """GroupNorm (+ SiLU) from the producers' accumulated fixed-point column sums: ONE launch, no statistics pass.  Returns None
    when the channel layout does not slice into whole groups (SEER_ENOSYS).  stats_out [batch, groups, 2] fp32: also receives
    (sum, sum of squares) per (batch element, group) for the backward pass."""
    dt = _req16(x1, "x1")
    if x2 is not None:
        _req16(x2, "x2", x1)
    rows = x1.shape[0] // batch
    C2 = 0 if x2 is None else x2.shape[1]
    assert fx1.C == x1.shape[1] and fx1.buf.shape[1] == batch and (x2 is None or (fx2.C == C2 and fx2.buf.shape[1] == batch))
    if out is None:
        out = torch.empty((x1.shape[0], x1.shape[1] + C2), device=x1.device, dtype=x1.dtype)
    _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    if stats_out is not None:
        _req(stats_out, torch.float32, "stats_out")
        assert stats_out.shape == (batch, groups, 2) and stats_out.is_contiguous()
    rc = _lib.load().seer_groupnorm_apply_fx_dt(_p(x1), x1.shape[1], _p(x2), C2, _p(fx1.buf), fx1.reps,
                                                _p(fx2.buf) if x2 is not None else None, fx2.reps if x2 is not None else 0, batch, rows,
                                                groups, float(count), float(eps), _p(gamma), _p(beta), int(silu), _p(out),
                                                _p(stats_out), dt, _stream())
    if rc == _lib.SEER_ENOSYS:
        return None
    check(rc, "seer_groupnorm_apply_fx")
    return out


def groupnorm_stats_fx(x: torch.Tensor, batch: int, arena: Optional[FxArena] = None) -> ColSumsFx:
    """accumulated fixed-point (sum, sum of squares) per (batch element, column) FROM THE ACTIVATIONS x [batch * rows, C]: exact
    integer sums (every element rounded on its own), so shards of the rows can be added in any order.  The slot comes out of
    `arena` (zeroed at the head of the evaluation) or is a fresh zeroed tensor."""
    dt = _req16(x, "x")
    assert x.dim() == 2 and x.is_contiguous() and x.shape[0] % batch == 0
    C_ = x.shape[1]
    buf = arena.take(1, batch, C_) if arena is not None else None
    if buf is None:
        buf = torch.zeros((1, batch, 2, C_), device=x.device, dtype=torch.int64)
    check(_lib.load().seer_groupnorm_stats_fx(_p(x), C_, batch, x.shape[0] // batch, _p(buf), dt, _stream()), "seer_groupnorm_stats_fx")
    return ColSumsFx(buf, C_)


def groupnorm_stats_from_fx(fx1: ColSumsFx, fx2: Optional[ColSumsFx], batch: int, groups: int, stats: torch.Tensor) -> torch.Tensor:
    """stats[b][g] = (sum, sumsq) from accumulated column sums (torch ops; the fall-back of a layout groupnorm_apply_fx refuses)."""
    v = fx1.totals() if fx2 is None else torch.cat([fx1.totals(), fx2.totals()], dim=1)
    stats.copy_(v.view(batch, groups, -1, 2).sum(dim=2).to(torch.float32))
    return stats


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    dt = _req16(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), device=x.device, dtype=x.dtype)
    _req16(out, "out", x)
    check(_lib.load().seer_layernorm_dt(_p(x), x.shape[0], x.shape[1], x.stride(0), _p(gamma), _p(beta), float(eps),
                                        _p(out), out.stride(0), dt, _stream()), "seer_layernorm")
    return out


FF_FUSED_C, FF_FUSED_ROWS = 320, 96
# The engine takes the fused launch where its workgroups fill the chip: a workgroup owns 96 rows for the whole launch, so a launch
# takes rounds x 74 us, rounds = ceil(workgroups / CUs), whatever the last round holds -- from three quarters of the slots up it is
# ahead of the launches it replaces (scripts/lab_ff_fused.py: 24 576 rows 89 against 103 us, 18 432 rows 80 / 86, 12 288 rows 73 / 66,
# 36 864 rows 151 / 149, 98 304 rows 303 / 399).  The choice is made per CALL, on the rows the call holds: a CFG pair evaluated as
# one batch (24 576 rows) takes the fused launch at the 320-channel level, the same pair evaluated as two calls
# (ddim_video.py:205-207's other branch; a CFG half per rank: 12 288 rows each) takes layernorm + the two GEMMs -- the two branches
# run different kernels there and agree to the bf16 tolerance, not bit for bit (tests/test_gpu_unet.py::test_unbatched_cfg_branch).
FF_FUSED_MIN_ROWS = 18432
FF_FUSED_MIN_FILL = 0.74
_n_cu: dict = {}


def device_cus(device=None) -> int:
    """compute units of the device the launch goes to (256 on MI355X); cached per device"""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    n = _n_cu.get(idx)
    if n is None:
        n = _n_cu[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return n


def ff_fused_pays(rows: int, n_cu: Optional[int] = None) -> bool:
    if rows < FF_FUSED_MIN_ROWS:
        return False
    if n_cu is None:
        n_cu = device_cus()
    wgs = -(-rows // FF_FUSED_ROWS)
    return wgs / (n_cu * -(-wgs // n_cu)) >= FF_FUSED_MIN_FILL


def ff_fused_pack(w1: torch.Tensor, wcat: torch.Tensor):
    """The two weight matrices of ff_fused in the kernel's fragment order (seer_ff_fused_pack_w1 / _wcat): w1 [2560, 320] bf16 in the
    interleaved GEGLU row order, wcat [320, 1600] bf16 = [Wp | Wp W2].  Once per model."""
    _req16(w1, "w1"); _req16(wcat, "wcat", w1)          # (the pack kernels move 16-bit words: bf16 and fp16 alike)
    assert w1.shape == (8 * FF_FUSED_C, FF_FUSED_C) and w1.is_contiguous() and wcat.shape == (FF_FUSED_C, 5 * FF_FUSED_C) and wcat.is_contiguous()
    w1f, wcf = torch.empty_like(w1), torch.empty_like(wcat)
    check(_lib.load().seer_ff_fused_pack_w1(_p(w1), _p(w1f), _stream()), "seer_ff_fused_pack_w1")
    check(_lib.load().seer_ff_fused_pack_wcat(_p(wcat), _p(wcf), _stream()), "seer_ff_fused_pack_wcat")
    return w1f, wcf


def ff_fused(h: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w1f: torch.Tensor, b1: torch.Tensor,
             wcf: torch.Tensor, bcat: torch.Tensor, *, eps: float = 1e-5, out: Optional[torch.Tensor] = None,
             colsum_batch=0, pre=None) -> Optional[torch.Tensor]:
    """y = x + [Wp | Wp W2] [h | GEGLU(LayerNorm(h) W1^T + b1)] + bcat as ONE launch (seer_ff_fused_c320): the feed-forward of a
    transformer block and the transformer's proj_out with both residual adds, at the 320-channel level.  w1f, wcf from
    ff_fused_pack; b1 in the interleaved GEGLU row order, bcat = Wp b2 + bp.  colsum_batch as in gemm(): (B, arena) -> out.colsums =
    the ColSumsFx of y, B -> the per-tile ColSums (96-row tiles; only where no tile straddles two batch elements).
    pre = (a, wof, bo): the rows the launch reads as h are h + a Wo^T + bo -- the attention's to_out projection and its residual, in
    the same launch (seer_ff_fused_c320_pre; wof = rowchain_pack(Wo)); h itself is not written.  Returns None (nothing launched) when
    the shape is not the kernel's: C = 320."""
    M, Cc = h.shape
    if Cc != FF_FUSED_C or M == 0:
        return None
    dt = _req16(h, "h"); _req16(x, "x", h); _req16(w1f, "w1f", h); _req16(wcf, "wcf", h)
    assert x.shape == h.shape and h.stride(1) == 1 and x.stride(1) == 1
    assert w1f.numel() == 8 * Cc * Cc and w1f.is_contiguous() and wcf.numel() == 5 * Cc * Cc and wcf.is_contiguous()
    for t, n in ((gamma, "gamma"), (beta, "beta"), (b1, "b1"), (bcat, "bcat")):
        _req(t, torch.float32, n)
    if out is None:
        out = torch.empty((M, Cc), device=h.device, dtype=h.dtype)
    fx, fx_rows, cs, tiles = None, 0, None, None
    B, arena = colsum_batch if isinstance(colsum_batch, tuple) else (colsum_batch, None)
    if B > 0 and M % B == 0:
        if arena is not None and (M // B) % 16 == 0 and M // B >= FF_FUSED_ROWS:
            fx = arena.take(8, B, Cc)
        if fx is not None:
            fx_rows, cs = M // B, ColSumsFx(fx, Cc)
        elif (M // B) % FF_FUSED_ROWS == 0:
            tiles = torch.empty((1, M // FF_FUSED_ROWS, Cc, 2), device=h.device, dtype=torch.float32)
            cs = ColSums(tiles, Cc, 1, M // FF_FUSED_ROWS)
    pa, lda, pw, pb = None, 0, None, None
    if pre is not None:
        a, wof, bo = pre
        _req16(a, "pre a", h); _req16(wof, "pre wof", h); _req(bo, torch.float32, "pre bo")
        assert a.shape == h.shape and a.stride(1) == 1 and wof.numel() == Cc * Cc and wof.is_contiguous() and bo.numel() == Cc
        pa, lda, pw, pb = _p(a), a.stride(0), _p(wof), _p(bo)
    check(_lib.load().seer_ff_fused_c320_pre(pa, lda, pw, pb, _p(h), h.stride(0), _p(x), x.stride(0), _p(out), out.stride(0), M, _p(gamma),
                                             _p(beta), float(eps), _p(w1f), _p(b1), _p(wcf), _p(bcat),
                                             fx.data_ptr() if fx is not None else None, fx_rows, fx.shape[0] if fx is not None else 0,
                                             _p(tiles) if tiles is not None else None, dt, _stream()), "seer_ff_fused_c320")
    out.colsums = cs
    out.rowstats = None
    return out


ROWCHAIN_C, ROWCHAIN_ROWS = 320, 96


def rowchain_pack(w: torch.Tensor) -> torch.Tensor:
    """w [n * 320, 320] (16-bit) -> the n matrices in the fragment order of rowchain() (seer_rowchain_pack).  Once per model."""
    _req16(w, "w")
    assert w.dim() == 2 and w.shape[1] == ROWCHAIN_C and w.shape[0] % ROWCHAIN_C == 0 and w.is_contiguous()
    out = torch.empty_like(w)
    check(_lib.load().seer_rowchain_pack(_p(w), w.stride(0), w.shape[0] // ROWCHAIN_C, _p(out), _stream()), "seer_rowchain_pack")
    return out


def rowchain_pays(rows: int, n_cu: Optional[int] = None, products: int = 4) -> bool:
    """a workgroup owns 96 rows for the whole launch (as ff_fused), so a launch takes rounds x ~30 us (four products) / ~20 us (two)
    whatever its rows: ahead of the launches it replaces where its workgroups fill the chip's rounds (scripts/lab_rowchain.py,
    profiles/r06_lab_rowchain.log: GroupNorm -> proj_in -> norm1 -> q|k|v 24 576 rows 35.9 against 54.4 us, 12 288 rows 31.6 / 35.0,
    6 144 rows 30.7 / 23.3; to_out + residual -> norm2 -> to_q 24.5 / 27.3, 21.2 / 19.3, 20.5 / 15.1); 28 672 rows (config 2 at 14
    frames) are 299 workgroups -- a second round for 43 of them: not taken"""
    if n_cu is None:
        n_cu = device_cus()
    wgs = -(-rows // ROWCHAIN_ROWS)
    fill = wgs / (n_cu * -(-wgs // n_cu))
    if products >= 4 and wgs <= n_cu:
        return rows >= 12288          # one round: ahead from half the chip's workgroups up
    return rows >= 18432 and fill >= 0.74      # several rounds cost rounds x ~33 us whatever the last one holds (as ff_fused_pays)


def rowchain(inp: torch.Tensor, w1f: torch.Tensor, *, b1: Optional[torch.Tensor] = None, gn=None, res: Optional[torch.Tensor] = None,
             h_out=True, ln=None, w2f: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, col_scale=None, rotary=None):
    """h = [GroupNorm(inp)] W1^T + b1 [+ res];  out = LayerNorm(h) [W2_0 | ...]^T  as ONE launch at 320 channels (seer_rowchain_c320).
    gn = (stats [B, G, 2] fp32 -- or the ColSumsFx the producer of inp accumulated: no statistics launch --, count, eps, gamma, beta,
    rows_per_batch[, groups]); ln = (gamma, beta, eps); w1f / w2f from rowchain_pack;
    h_out: True = a new tensor, a tensor = write there (may be `res`), False = h is not stored; col_scale = (factor, thirds);
    rotary = (table, tokens_per_batch, pos_offset, head_dim, rot_dim, thirds).  Returns (h or None, out or None), or None when the
    launch is refused (SEER_ENOSYS: a 96-row tile would straddle two batch elements of the GroupNorm)."""
    dt = _req16(inp, "inp"); _req16(w1f, "w1f", inp)
    M, Cc = inp.shape
    assert Cc == ROWCHAIN_C and inp.stride(1) == 1 and w1f.numel() == Cc * Cc and w1f.is_contiguous()
    d = _lib.RowChainDesc()
    d.inp, d.ld_in, d.M, d.dtype = _p(inp), inp.stride(0), M, dt
    d.w1f = _p(w1f)
    if b1 is not None:
        _req(b1, torch.float32, "b1"); d.b1 = _p(b1)
    if gn is not None:
        stats, count, eps, gamma, beta, rows_pb = gn[:6]
        _req(gamma, torch.float32, "gn gamma"); _req(beta, torch.float32, "gn beta")
        assert M % rows_pb == 0
        if isinstance(stats, ColSumsFx):
            assert stats.C == Cc and stats.buf.shape[1] == M // rows_pb and len(gn) > 6
            d.gn_fx, d.gn_fx_reps, d.groups = _p(stats.buf), stats.reps, int(gn[6])
        else:
            _req(stats, torch.float32, "gn stats")
            assert stats.is_contiguous() and stats.dim() == 3 and stats.shape[2] == 2 and stats.shape[0] == M // rows_pb
            d.gn_stats, d.groups = _p(stats), stats.shape[1]
        d.gn_count, d.gn_eps, d.gn_gamma, d.gn_beta = float(count), float(eps), _p(gamma), _p(beta)
        d.rows_per_batch = rows_pb
    if res is not None:
        _req16(res, "res", inp)
        assert res.shape == (M, Cc) and res.stride(1) == 1
        d.res, d.ldr = _p(res), res.stride(0)
    h = None
    if h_out is not False:
        h = torch.empty((M, Cc), device=inp.device, dtype=inp.dtype) if h_out is True else h_out
        _req16(h, "h", inp)
        assert h.shape == (M, Cc) and h.stride(1) == 1
        d.h, d.ldh = _p(h), h.stride(0)
    if ln is not None:
        g, b, eps = ln
        _req(g, torch.float32, "ln gamma"); _req(b, torch.float32, "ln beta")
        d.ln_gamma, d.ln_beta, d.ln_eps = _p(g), _p(b), float(eps)
    o = None
    if w2f is not None:
        _req16(w2f, "w2f", inp)
        n2 = w2f.numel() // (Cc * Cc)
        assert w2f.is_contiguous() and n2 * Cc * Cc == w2f.numel() and 1 <= n2 <= 3
        o = torch.empty((M, n2 * Cc), device=inp.device, dtype=inp.dtype) if out is None else out
        _req16(o, "out", inp)
        assert o.shape == (M, n2 * Cc) and o.stride(1) == 1
        d.w2f, d.n2, d.out, d.ldo = _p(w2f), n2, _p(o), o.stride(0)
        if col_scale is not None:
            d.col_scale, d.scale_thirds = float(col_scale[0]), int(col_scale[1])
        if rotary is not None:
            table, tpb, pos_off, hd, rd, thirds = rotary
            _req(table, torch.float32, "rotary table")
            assert table.shape[0] >= tpb + pos_off and table.shape[1] * 2 == rd
            d.rot_table, d.rot_tokens_per_batch, d.rot_pos_offset, d.rot_head_dim, d.rot_dim, d.rot_thirds = _p(table), tpb, pos_off, hd, rd, thirds
    rc = _lib.load().seer_rowchain_c320(C.byref(d), _stream())
    if rc == _lib.SEER_ENOSYS:
        return None
    check(rc, "seer_rowchain_c320")
    if h is not None:
        h.colsums = None
        h.rowstats = None
    return h, o


def softmax_rows(x: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None, dtype=None) -> torch.Tensor:
    """softmax(scale * x) over the last dim; x bf16 / fp16 / fp32 -> `dtype` (bf16 unless out / dtype say fp16)."""
    assert x.dtype in (bf16, f16, torch.float32) and x.is_cuda
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype if dtype is not None else (x.dtype if x.dtype != torch.float32 else bf16))
    ldy = out.reshape(-1, out.shape[-1]).stride(0)        # `out` may be wider than x (zero-padded contraction of the next GEMM)
    dt = _req16(out, "out", None if x.dtype == torch.float32 else x)
    assert out.shape[-1] >= x2.shape[1] and out.numel() // out.shape[-1] == x2.shape[0]
    check(_lib.load().seer_softmax_rows_dt(_p(x2), int(x.dtype == torch.float32), x2.shape[0], x2.shape[1], x2.stride(0),
                                           float(scale), _p(out), ldy, dt, _stream()), "seer_softmax_rows")
    return out


def conv1x1_nchw(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """pointwise conv on NCHW fp32 (VAE post_quant_conv); w fp32 [Cout, Cin]."""
    _req(x, torch.float32, "x"); _req(w, torch.float32, "w")
    x = x.contiguous()
    N, Cin = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cin)
    Cout = w.shape[0]
    y = torch.empty((N, Cout) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
    check(_lib.load().seer_conv1x1_nchw_f32(_p(x), N, Cin, Cout, HW, _p(w), _p(bias), _p(y), _stream()),
          "seer_conv1x1_nchw_f32")
    return y


# ------------------------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float) -> torch.Tensor:
    _req(t, torch.int64, "timestep")
    out = torch.empty((t.numel(), dim), device=t.device, dtype=torch.float32)
    check(_lib.load().seer_timestep_embedding(_p(t), t.numel(), dim, int(flip_sin_to_cos), float(freq_shift), _p(out),
                                              _stream()), "seer_timestep_embedding")
    return out


def linear_smallm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, silu_in=False,
                  silu_out=False) -> torch.Tensor:
    _req(x, torch.float32, "x")
    dt = _req16(w, "w")
    assert x.is_contiguous() and w.is_contiguous()
    B, K = x.shape
    N = w.shape[0]
    y = torch.empty((B, N), device=x.device, dtype=torch.float32)
    check(_lib.load().seer_linear_smallm_dt(_p(x), B, K, _p(w), _p(bias), N, int(silu_in), int(silu_out), _p(y), dt,
                                            _stream()), "seer_linear_smallm")
    return y


def conv_in(x: torch.Tensor, w_khwc: torch.Tensor, bias: torch.Tensor, dtype=bf16) -> torch.Tensor:
    """x [B, Cin, F, H, W] fp32 -> [B*F*H*W, Cout] bf16 (or fp16) ; w_khwc fp32 [3,3,Cin,Cout]."""
    _req(x, torch.float32, "x"); _req(w_khwc, torch.float32, "w")
    assert x.is_contiguous()
    B, Cin, F, H, W = x.shape
    Cout = w_khwc.shape[-1]
    y = torch.empty((B * F * H * W, Cout), device=x.device, dtype=dtype)
    dt = _req16(y, "y")
    check(_lib.load().seer_conv_in_dt(_p(x), B, Cin, F, H, W, _p(w_khwc), _p(bias), Cout, _p(y), dt, _stream()),
          "seer_conv_in")
    return y


def conv_out(x: torch.Tensor, w_ohwc: torch.Tensor, bias: torch.Tensor, B: int, F: int, H: int, W: int) -> torch.Tensor:
    """x [B*F*H*W, C0] bf16 -> [B, Cout, F, H, W] fp32 ; w fp32 [Cout,3,3,C0] (direct kernel) or, for Cout % 4 == 0,
    bf16 [Cout, 9*C0] in conv3x3 packing: implicit-GEMM on the MFMA path, batched over B with a transposed fp32 store
    (the [Cout, F*H*W] planes of NCFHW are C^T of the per-sample GEMM)."""
    dt = _req16(x, "x")
    Cout = w_ohwc.shape[0]
    y = torch.empty((B, Cout, F, H, W), device=x.device, dtype=torch.float32)
    if w_ohwc.dtype in (bf16, f16):
        _req16(w_ohwc, "w", x)
        C0 = x.shape[1]
        assert w_ohwc.dim() == 2 and w_ohwc.shape[1] == 9 * C0 and Cout % 4 == 0 and x.is_contiguous()
        M = F * H * W
        d = GemmDesc()
        d.A, d.W, d.C = _p(x), _p(w_ohwc), _p(y)
        d.M, d.N, d.K, d.K1 = M, Cout, 9 * C0, 9 * C0
        d.ldc = M
        _req(bias, torch.float32, "bias")
        d.bias = _p(bias)
        d.mode = _lib.SEER_GEMM_CONV3X3
        d.epilogue = _lib.SEER_EPI_TRANS_OUT | _lib.SEER_EPI_OUT_F32 | (_lib.SEER_EPI_F16 if dt else 0)
        d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride, d.upsample = H, W, C0, H, W, 1, 0
        d.batch, d.strideA, d.strideW, d.strideC = B, M * C0, 0, Cout * M
        d.splits = 1
        _launch_gemm(d, x.device, "seer_gemm_bf16(conv_out)")
        return y
    _req(w_ohwc, torch.float32, "w")
    check(_lib.load().seer_conv_out_dt(_p(x), B, x.shape[1], F, H, W, _p(w_ohwc), _p(bias), Cout, _p(y), dt, _stream()),
          "seer_conv_out")
    return y


def cast_bf16(x: torch.Tensor, dtype=bf16) -> torch.Tensor:
    """fp32 -> bf16 (or, dtype=torch.float16, IEEE half: the fp16 engine's context)"""
    _req(x, torch.float32, "x")
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(_lib.load().seer_cast_f32_dt(_p(x), x.numel(), _p(y), _req16(y, "y"), _stream()), "seer_cast_f32_dt")
    return y


def nchw_to_nhwc_bf16(x: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] fp32 -> [N*H*W, C] bf16"""
    _req(x, torch.float32, "x")
    x = x.contiguous()
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    y = torch.empty((N * HW, Cc), device=x.device, dtype=bf16)
    check(_lib.load().seer_nchw_f32_to_nhwc_bf16(_p(x), N, Cc, HW, _p(y), _stream()), "seer_nchw_f32_to_nhwc_bf16")
    return y


def nhwc_to_nchw_f32(x: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
    _req(x, bf16, "x")
    Cc = x.shape[1]
    y = torch.empty((N, Cc, H, W), device=x.device, dtype=torch.float32)
    check(_lib.load().seer_nhwc_bf16_to_nchw_f32(_p(x), N, Cc, H * W, _p(y), _stream()), "seer_nhwc_bf16_to_nchw_f32")
    return y


def cfg_ddim_step(eps: torch.Tensor, x: torch.Tensor, coef: torch.Tensor, index: int, *, cfg: bool, scale: float,
                  cond_f: int, noise: Optional[torch.Tensor] = None, want_pred_x0=True):
    """eps [2b or b, C, F_total, h, w] fp32 ; x [b, C, F_pred, h, w] fp32 -> (x_prev, pred_x0)."""
    _req(eps, torch.float32, "eps"); _req(x, torch.float32, "x"); _req(coef, torch.float32, "coef")
    assert eps.is_contiguous() and x.is_contiguous()
    b, Cc, Fp, h, w = x.shape
    Ft = eps.shape[2]
    assert Ft == Fp + cond_f and eps.shape[0] == (2 * b if cfg else b)
    x_prev = torch.empty_like(x)
    pred = torch.empty_like(x) if want_pred_x0 else None
    check(_lib.load().seer_cfg_ddim_step(_p(eps), int(cfg), b, Cc, Ft, cond_f, h * w, float(scale), _p(coef), index,
                                         _p(x), _p(noise), _p(x_prev), _p(pred), _stream()), "seer_cfg_ddim_step")
    return x_prev, pred


def ddim_step_begin(x0_emb: Optional[torch.Tensor], x: torch.Tensor, t_table: torch.Tensor, step: torch.Tensor, reps: int,
                    sample: torch.Tensor, t_out: torch.Tensor) -> None:
    """first kernel of a captured sampler step: sample[reps*b, C, f1+Fp, h, w] = cat([x0_emb, x], 2) x reps,
    t_out[:] = t_table[step[0]], step[1] = step[0] (include/seer_hip.h, seer_ddim_step_begin)"""
    _req(x, torch.float32, "x"); _req(t_table, torch.int64, "t_table"); _req(step, torch.int32, "step")
    b, Cc, Fp, h, w = x.shape
    f1 = 0 if x0_emb is None else x0_emb.shape[2]
    if x0_emb is not None:
        _req(x0_emb, torch.float32, "x0_emb")
        assert x0_emb.is_contiguous() and x0_emb.shape[:2] == x.shape[:2] and x0_emb.shape[3:] == x.shape[3:]
    assert x.is_contiguous() and sample.is_contiguous() and sample.shape == (reps * b, Cc, f1 + Fp, h, w) and step.numel() >= 2
    assert sample.dtype == torch.float32 and t_out.dtype == torch.int64 and t_out.numel() == reps * b
    check(_lib.load().seer_ddim_step_begin(_p(x0_emb), _p(x), b, reps, Cc, f1, Fp, h * w, _p(t_table), _p(step), _p(sample),
                                           _p(t_out), _stream()), "seer_ddim_step_begin")


def cfg_ddim_step_dev(eps: torch.Tensor, x: torch.Tensor, coef: torch.Tensor, step: torch.Tensor, *, cfg: bool, scale: float,
                      cond_f: int, x_prev: torch.Tensor, pred_x0: Optional[torch.Tensor], noise: Optional[torch.Tensor] = None):
    """last kernel of a captured sampler step: cfg_ddim_step with index = step[1], then step[0] = index - 1; x_prev may be x"""
    _req(eps, torch.float32, "eps"); _req(x, torch.float32, "x"); _req(coef, torch.float32, "coef"); _req(step, torch.int32, "step")
    assert eps.is_contiguous() and x.is_contiguous() and x_prev.is_contiguous() and x_prev.shape == x.shape
    b, Cc, Fp, h, w = x.shape
    Ft = eps.shape[2]
    assert Ft == Fp + cond_f and eps.shape[0] == (2 * b if cfg else b)
    check(_lib.load().seer_cfg_ddim_step_dev(_p(eps), int(cfg), b, Cc, Ft, cond_f, h * w, float(scale), _p(coef), _p(step), _p(x),
                                             _p(noise), _p(x_prev), _p(pred_x0), _stream()), "seer_cfg_ddim_step_dev")


def clamp01_(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    check(_lib.load().seer_clamp01(_p(x), x.numel(), _stream()), "seer_clamp01")
    return x


def gaussian_sample(moments: torch.Tensor, noise: Optional[torch.Tensor]) -> torch.Tensor:
    """moments fp32 [N, 2C, H, W] (mean | logvar) -> mean + exp(0.5 clamp(logvar, -30, 20)) * noise (noise None: the mean)."""
    _req(moments, torch.float32, "moments")
    assert moments.is_contiguous() and moments.dim() == 4 and moments.shape[1] % 2 == 0
    N, C2, H, W = moments.shape
    out = torch.empty((N, C2 // 2, H, W), device=moments.device, dtype=torch.float32)
    if noise is not None:
        _req(noise, torch.float32, "noise")
        assert noise.shape == out.shape and noise.is_contiguous()
    check(_lib.load().seer_gaussian_sample(_p(moments), N, C2 // 2, H * W, _p(noise), _p(out), _stream()),
          "seer_gaussian_sample")
    return out
