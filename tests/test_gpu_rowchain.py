"""seer_rowchain_c320 (csrc/rowchain.hip): the row-local chains in front of the attention launches of a 320-channel transformer block as
ONE launch -- GroupNorm -> proj_in -> norm1 -> to_q | to_k | to_v (+ rotary, + the q prescale), and attn1.to_out + residual -> norm2 ->
attn2.to_q (seer/models/attention.py:129-145, 231-240, 308-322, 649-651) -- against the fp32 formula and against the launches it replaces."""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu
bf16, f16 = torch.bfloat16, torch.float16
C = 320


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _rotary_ref(y, M, heads, d, rot, freqs, T, off=0):
    pos = (torch.arange(M, device=y.device) % T + off).float()
    ang = pos[:, None] * freqs[None, :]
    v = y.reshape(M, heads, d).clone()
    x0, x1 = v[:, :, 0:rot:2].clone(), v[:, :, 1:rot:2].clone()
    v[:, :, 0:rot:2] = x0 * ang.cos()[:, None, :] - x1 * ang.sin()[:, None, :]
    v[:, :, 1:rot:2] = x1 * ang.cos()[:, None, :] + x0 * ang.sin()[:, None, :]
    return v.reshape(M, heads * d)


@pytest.mark.parametrize("dt,B,rows_pb,rotary", [(bf16, 2, 12288, False), (bf16, 2, 1536, True), (f16, 2, 3072, True), (bf16, 1, 96, False),
                                                  (bf16, 3, 1000, True), (f16, 8, 200, False)])      # tiles that span two batch elements
def test_groupnorm_proj_in_layernorm_qkv(device, dt, B, rows_pb, rotary):
    from seervideoldm_amd import ops
    M, G = B * rows_pb, 32
    x = (_rand((M, C), device, 1, 1.5) + 0.3).to(dt)
    gg, gb = _rand((C,), device, 2) * 0.2 + 1.0, _rand((C,), device, 3) * 0.2
    wp, bp = _rand((C, C), device, 4, C ** -0.5).to(dt), _rand((C,), device, 5) * 0.1
    lg, lb = _rand((C,), device, 6) * 0.2 + 1.0, _rand((C,), device, 7) * 0.2
    wqkv = _rand((3 * C, C), device, 8, C ** -0.5).to(dt)
    stats = torch.zeros((B, G, 2), device=device)
    ops.groupnorm_stats(x, None, B, G, stats)
    count = rows_pb * (C // G)
    heads, d, rot, T = 8, 40, 32, rows_pb
    freqs = (10000.0 ** (-torch.arange(0, rot, 2, dtype=torch.float32) / rot)).to(device)
    table = ops.rotary_table(freqs, T) if rotary else None
    sc = ops.qk_prescale(d)
    got = ops.rowchain(x, ops.rowchain_pack(wp), b1=bp, gn=(stats, count, 1e-6, gg, gb, rows_pb), ln=(lg, lb, 1e-5),
                       w2f=ops.rowchain_pack(wqkv), col_scale=(sc, 1), rotary=(table, T, 0, d, rot, 2) if rotary else None)
    assert got is not None
    h, qkv = got
    assert h.dtype == dt and qkv.dtype == dt and h.shape == (M, C) and qkv.shape == (M, 3 * C)
    # fp32 formula on the rounded operands; the stored h is what the LayerNorm sees
    xn = Fn.group_norm(x.float().reshape(B, rows_pb, C).permute(0, 2, 1), G, gg, gb, 1e-6).permute(0, 2, 1).reshape(M, C)
    h_ref = xn @ wp.float().t() + bp
    tol = 8e-3 if dt == bf16 else 1.2e-3
    assert _rel(h, h_ref) < tol, _rel(h, h_ref)
    y = Fn.layer_norm(h.float(), (C,), lg, lb, 1e-5) @ wqkv.float().t()
    if rotary:
        y[:, :C] = _rotary_ref(y[:, :C], M, heads, d, rot, freqs, T)
        y[:, C:2 * C] = _rotary_ref(y[:, C:2 * C], M, heads, d, rot, freqs, T)
    y[:, :C] *= sc
    assert _rel(qkv, y) < tol, _rel(qkv, y)
    # ... and the launches it replaces: GroupNorm apply, proj_in, layernorm, the q|k|v projection -- same roundings at the same places
    xa = ops.groupnorm_apply(x, None, B, G, stats, count, 1e-6, gg, gb, False)
    h3 = ops.gemm(xa, wp, bias=bp)
    q3 = ops.gemm(ops.layernorm(h3, lg, lb), wqkv, rotary=(table, T, 0, d, rot, 2 * C) if rotary else None, col_scale=(sc, C))
    assert _rel(h, h3) < tol / 2 and _rel(qkv, q3) < tol, (_rel(h, h3), _rel(qkv, q3))
    # deterministic
    h2, qkv2 = ops.rowchain(x, ops.rowchain_pack(wp), b1=bp, gn=(stats, count, 1e-6, gg, gb, rows_pb), ln=(lg, lb, 1e-5),
                            w2f=ops.rowchain_pack(wqkv), col_scale=(sc, 1), rotary=(table, T, 0, d, rot, 2) if rotary else None)
    assert torch.equal(h, h2) and torch.equal(qkv, qkv2)


@pytest.mark.parametrize("dt,M", [(bf16, 24576), (f16, 6144), (bf16, 1000)])
def test_to_out_residual_layernorm_q_in_place(device, dt, M):
    """attn1.to_out + residual (written over the residual stream) -> norm2 -> attn2.to_q with the q prescale; M = 1000: a ragged last
    tile (no GroupNorm on this chain: no batch-element rule)"""
    from seervideoldm_amd import ops
    a = _rand((M, C), device, 1).to(dt)
    h0 = _rand((M, C), device, 2).to(dt)
    wo, bo = _rand((C, C), device, 3, C ** -0.5).to(dt), _rand((C,), device, 4) * 0.1
    lg, lb = _rand((C,), device, 5) * 0.2 + 1.0, _rand((C,), device, 6) * 0.2
    wq = _rand((C, C), device, 7, C ** -0.5).to(dt)
    sc = ops.qk_prescale(40)
    h = h0.clone()
    got = ops.rowchain(a, ops.rowchain_pack(wo), b1=bo, res=h, h_out=h, ln=(lg, lb, 1e-5), w2f=ops.rowchain_pack(wq), col_scale=(sc, 1))
    assert got is not None and got[0] is h
    q = got[1]
    h_ref = a.float() @ wo.float().t() + bo + h0.float()
    tol = 8e-3 if dt == bf16 else 1.2e-3
    assert _rel(h, h_ref) < tol
    q_ref = Fn.layer_norm(h.float(), (C,), lg, lb, 1e-5) @ wq.float().t() * sc
    assert q.shape == (M, C) and _rel(q, q_ref) < tol
    # first product only (no second matrix): h alone
    h1 = ops.rowchain(a, ops.rowchain_pack(wo), b1=bo, res=h0)[0]
    assert torch.equal(h1, h)


def test_rowchain_refuses_a_tile_across_three_batch_elements(device):
    from seervideoldm_amd import ops
    x = _rand((4 * 40, C), device, 1).to(bf16)
    w = ops.rowchain_pack(_rand((C, C), device, 2, 0.05).to(bf16))
    stats = torch.zeros((4, 32, 2), device=device)
    ops.groupnorm_stats(x, None, 4, 32, stats)
    g = _rand((C,), device, 3)
    assert ops.rowchain(x, w, gn=(stats, 40 * 10, 1e-6, g, g, 40)) is None


def test_groupnorm_from_the_producers_accumulated_sums(device):
    """the chain reads the fixed-point column sums its input's producer accumulated (seer_gemm_desc::colsum_fx): no statistics launch
    in front; same result as from the (sum, sum of squares) per group of a statistics pass up to the statistics' rounding"""
    from seervideoldm_amd import ops
    B, rows_pb, G = 2, 12288, 32
    M = B * rows_pb
    a = _rand((M, 320), device, 1).to(bf16)
    wprod = _rand((C, 320), device, 2, 320 ** -0.5).to(bf16)
    arena = ops.FxArena(device, 1 << 18)
    x = ops.gemm(a, wprod, bias=_rand((C,), device, 3), colsum_batch=(B, arena))
    assert isinstance(x.colsums, ops.ColSumsFx)
    gg, gb = _rand((C,), device, 4) * 0.2 + 1.0, _rand((C,), device, 5) * 0.2
    wp = ops.rowchain_pack(_rand((C, C), device, 6, C ** -0.5).to(bf16))
    wq = ops.rowchain_pack(_rand((3 * C, C), device, 7, C ** -0.5).to(bf16))
    lg, lb = _rand((C,), device, 8) * 0.2 + 1.0, _rand((C,), device, 9) * 0.2
    count = rows_pb * (C // G)
    h1, q1 = ops.rowchain(x, wp, gn=(x.colsums, count, 1e-6, gg, gb, rows_pb, G), ln=(lg, lb, 1e-5), w2f=wq)
    stats = torch.zeros((B, G, 2), device=device)
    ops.groupnorm_stats(x, None, B, G, stats)
    h2, q2 = ops.rowchain(x, wp, gn=(stats, count, 1e-6, gg, gb, rows_pb), ln=(lg, lb, 1e-5), w2f=wq)
    assert _rel(h1, h2) < 2e-3 and _rel(q1, q2) < 2e-3, (_rel(h1, h2), _rel(q1, q2))
    xn = Fn.group_norm(x.float().reshape(B, rows_pb, C).permute(0, 2, 1), G, gg, gb, 1e-6).permute(0, 2, 1).reshape(M, C)
    # (reference through the same packed weights is the other test's job; here: the normalisation itself)
    h3 = ops.rowchain(x, wp, gn=(x.colsums, count, 1e-6, gg, gb, rows_pb, G))[0]
    assert torch.equal(h3, h1)
