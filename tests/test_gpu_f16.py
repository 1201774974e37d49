"""The fp16-storage variants of the kernels the VAE runs on (SEER_EPI_F16 / SEER_DT_F16): the reference decodes and encodes in
fp32 (inference_img.py:118,168; ldm/modules/diffusionmodules/model.py:368-568), so the VAE path keeps 11 significand bits
instead of bf16's 8.  Each op against the fp32 formula on fp16-rounded inputs, with a bound ~8x tighter than the bf16 tests'."""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

f16 = torch.float16


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _close(got, ref, rtol=3e-3, atol=3e-3, what=""):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside tolerance, max err {err.max().item():.4g}"


@pytest.mark.parametrize("M,N,K", [(4096, 512, 512), (1024, 128, 1152), (300, 132, 192), (65536, 128, 128), (1024, 512, 4608)])
def test_gemm_f16(device, M, N, K):
    from seervideoldm_amd import ops
    a = _rand((M, K), device, 1).to(f16)
    w = _rand((N, K), device, 2, K ** -0.5).to(f16)
    bias = _rand((N,), device, 3)
    res = _rand((M, N), device, 4).to(f16)
    out = ops.gemm(a, w, bias=bias, residual=res)
    assert out.dtype == f16
    _close(out, a.float() @ w.float().t() + bias + res.float(), what=f"f16 gemm {M}x{N}x{K}")
    _close(ops.gemm(a, w, out_f32=True), a.float() @ w.float().t(), rtol=1e-3, atol=1e-3, what="f16 gemm, fp32 out")
    with pytest.raises(TypeError):
        ops.gemm(a, w.to(torch.bfloat16))


def test_gemm_batched_f16(device):
    from seervideoldm_amd import ops
    a = _rand((3, 256, 512), device, 1).to(f16)
    w = _rand((3, 192, 512), device, 2, 512 ** -0.5).to(f16)
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float())
    _close(ops.gemm_batched(a, w), ref, what="f16 batched")
    _close(ops.gemm_batched(a, w, trans_out=True), ref.transpose(1, 2), what="f16 batched, transposed store")
    _close(ops.gemm_batched(a, w, out_f32=True), ref, rtol=1e-3, atol=1e-3, what="f16 batched, fp32 out")


@pytest.mark.parametrize("n_img,H,W,Ci,Co,stride,up,pad_after", [
    (2, 32, 32, 128, 128, 1, False, False), (2, 16, 16, 512, 512, 1, False, False), (2, 16, 16, 256, 256, 1, True, False),
    (2, 32, 32, 128, 128, 2, False, True), (3, 6, 10, 64, 68, 1, False, False),
])
def test_conv3x3_f16(device, n_img, H, W, Ci, Co, stride, up, pad_after):
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    x = _rand((n_img, Ci, H, W), device, 1).to(f16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5).to(f16)
    bias = _rand((Co,), device, 3)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    out = ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, stride=stride, upsample=up, bias=bias, pad_after_only=pad_after)
    xin = x.float()
    if up:
        xin = Fn.interpolate(xin, scale_factor=2.0, mode="nearest")
    if pad_after:
        ref = Fn.conv2d(Fn.pad(xin, (0, 1, 0, 1)), w.float(), bias, stride=stride)
    else:
        ref = Fn.conv2d(xin, w.float(), bias, stride=stride, padding=1)
    assert out.dtype == f16
    _close(out, ref.permute(0, 2, 3, 1).reshape(-1, Co), what=f"f16 conv {Ci}->{Co}")


@pytest.mark.parametrize("B,rows,C,silu", [(2, 1024, 128, True), (1, 4096, 512, False), (3, 200, 256, True)])
def test_groupnorm_f16(device, B, rows, C, silu):
    from seervideoldm_amd import ops
    G = 32
    x = _rand((B * rows, C), device, 1, 3.0).to(f16)
    gamma, beta = _rand((C,), device, 2) + 1.0, _rand((C,), device, 3)
    stats = torch.zeros((B, G, 2), device=device, dtype=torch.float32)
    ops.groupnorm_stats(x, None, B, G, stats)
    y = ops.groupnorm_apply(x, None, B, G, stats, rows * (C // G), 1e-6, gamma, beta, silu)
    assert y.dtype == f16
    xr = x.float().reshape(B, rows, C).permute(0, 2, 1)
    ref = Fn.group_norm(xr, G, gamma, beta, 1e-6)
    if silu:
        ref = Fn.silu(ref)
    _close(y, ref.permute(0, 2, 1).reshape(B * rows, C), what="f16 groupnorm")


def test_softmax_and_boundary_convs_f16(device):
    from seervideoldm_amd import ops
    s = _rand((2, 64, 1024), device, 1, 4.0)
    p = ops.softmax_rows(s, 0.25, dtype=f16)
    assert p.dtype == f16
    _close(p, torch.softmax(s * 0.25, -1), rtol=2e-3, atol=1e-4, what="f16 softmax")
    # conv_in (fp32 NCFHW -> channels-last fp16) and the RGB conv_out (channels-last fp16 -> fp32 NCFHW)
    x = _rand((2, 4, 1, 16, 16), device, 2)
    w = _rand((128, 4, 3, 3), device, 3, 0.2)
    b = _rand((128,), device, 4)
    y = ops.conv_in(x, w.permute(2, 3, 1, 0).contiguous(), b, dtype=f16)
    ref = Fn.conv2d(x[:, :, 0], w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, 128)
    _close(y, ref, what="f16 conv_in")
    h = _rand((2 * 16 * 16, 128), device, 5).to(f16)
    wo = _rand((3, 128, 3, 3), device, 6, 0.05)
    bo = _rand((3,), device, 7)
    img = ops.conv_out(h, wo.permute(0, 2, 3, 1).contiguous(), bo, 2, 1, 16, 16)
    ref = Fn.conv2d(h.float().reshape(2, 16, 16, 128).permute(0, 3, 1, 2), wo, bo, padding=1)
    _close(img.reshape(2, 3, 16, 16), ref, rtol=1e-3, atol=1e-3, what="f16 conv_out")


# ---- the UNet engine on fp16 storage (every shipped yaml says mixed_precision: "fp16"): the epilogues, split-K, statistics, LayerNorm
# fold, attention and norm launches the VAE never needed, each against its fp32 formula on fp16-rounded inputs ---------------------------
def test_gemm_f16_geglu_rotary_colscale(device):
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import geglu_row_order
    M, K, C = 2048, 320, 320
    a = _rand((M, K), device, 1).to(f16)
    # GEGLU (interleaved row order, value * gelu(gate))
    w = _rand((8 * C, K), device, 2, K ** -0.5).to(f16)
    b = _rand((8 * C,), device, 3)
    order = geglu_row_order(4 * C).to(device)
    g = ops.gemm(a, w[order].contiguous(), bias=b[order].contiguous(), geglu=True)
    h = a.float() @ w.float().t() + b
    _close(g, h[:, :4 * C] * Fn.gelu(h[:, 4 * C:]), what="f16 GEGLU")
    assert g.dtype == f16
    # rotary + column scale on a fused q|k|v projection (d = 40: the first 32 channels of every q / k head rotate)
    heads, d, rot = 8, 40, 32
    wq = _rand((3 * C, K), device, 4, K ** -0.5).to(f16)
    freqs = (10000.0 ** (-torch.arange(0, rot, 2, dtype=torch.float32) / rot)).to(device)
    T = 512
    table = ops.rotary_table(freqs, T)
    sc = ops.qk_prescale(d)
    y = ops.gemm(a, wq, rotary=(table, T, 0, d, rot, 2 * C), col_scale=(sc, C))
    ref = a.float() @ wq.float().t()
    pos = (torch.arange(M, device=device) % T).float()
    ang = pos[:, None] * freqs[None, :]                             # [M, rot / 2]
    qk = ref[:, :2 * C].reshape(M, 2 * heads, d).clone()
    x0, x1 = qk[:, :, 0:rot:2].clone(), qk[:, :, 1:rot:2].clone()
    qk[:, :, 0:rot:2] = x0 * ang.cos()[:, None, :] - x1 * ang.sin()[:, None, :]
    qk[:, :, 1:rot:2] = x1 * ang.cos()[:, None, :] + x0 * ang.sin()[:, None, :]
    ref[:, :2 * C] = qk.reshape(M, 2 * C)
    ref[:, :C] *= sc
    _close(y, ref, what="f16 rotary + column scale")


@pytest.mark.parametrize("M,N,K,conv", [(1536, 1280, 5120, False), (384, 1280, 1280 * 9, True), (6144, 640, 640 * 9, True)])
def test_gemm_f16_splitk_and_statistics(device, M, N, K, conv):
    """the split-K launches of the deep levels (fp32 slices + the ordered reduce pass, whose epilogue stores fp16 here) with the
    accumulated fixed-point column sums of the GroupNorm that follows, and the one-launch apply that reads them"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    B = 2
    bias = _rand((N,), device, 3)
    arena = ops.FxArena(device, 1 << 20)
    if conv:
        side = {384: 4, 6144: 16}[M]
        n_img, Ci = M // (side * side), K // 9
        x = _rand((n_img, Ci, side, side), device, 1).to(f16)
        w = _rand((N, Ci, 3, 3), device, 2, K ** -0.5).to(f16)
        res = _rand((M, N), device, 4).to(f16)
        out = ops.conv3x3(x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous(), pack_conv3x3(w), n_img, side, side, bias=bias, residual=res,
                          colsum_batch=(B, arena))
        ref = Fn.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(M, N) + res.float()
    else:
        a = _rand((M, K), device, 1).to(f16)
        w = _rand((N, K), device, 2, K ** -0.5).to(f16)
        res = _rand((M, N), device, 4).to(f16)
        out = ops.gemm(a, w, bias=bias, residual=res, colsum_batch=(B, arena))
        ref = a.float() @ w.float().t() + bias + res.float()
    assert out.dtype == f16
    _close(out, ref, rtol=4e-3, atol=4e-3, what="f16 split-K")
    cs = out.colsums
    assert isinstance(cs, ops.ColSumsFx), "the launch must leave accumulated column sums"
    tot = cs.totals()                                                           # [B, N, 2] fp64 of the STORED values
    st = out.float().double().reshape(B, M // B, N)
    assert torch.allclose(tot[:, :, 0], st.sum(1), rtol=0, atol=1e-2) and torch.allclose(tot[:, :, 1], (st * st).sum(1), rtol=1e-6, atol=1e-2)
    G = 32
    gamma, beta = _rand((N,), device, 5) + 1.0, _rand((N,), device, 6)
    y = ops.groupnorm_apply_fx(out, None, cs, None, B, G, (M // B) * (N // G), 1e-5, gamma, beta, True)
    assert y is not None and y.dtype == f16
    r = Fn.silu(Fn.group_norm(out.float().reshape(B, M // B, N).permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(M, N)
    _close(y, r, what="f16 groupnorm from accumulated sums")


def test_layernorm_f16_and_fold(device):
    from seervideoldm_amd import ops
    M, C, N = 6144, 640, 1920
    x = _rand((M, C), device, 1, 2.0).to(f16)
    gamma, beta = _rand((C,), device, 2) + 1.0, _rand((C,), device, 3)
    y = ops.layernorm(x, gamma, beta)
    assert y.dtype == f16
    ref_ln = Fn.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    _close(y, ref_ln, what="f16 layernorm")
    # the fold: a producer leaves row statistics, the consumer normalises in its epilogue
    a = _rand((M, C), device, 4).to(f16)
    wp = _rand((C, C), device, 5, C ** -0.5).to(f16)
    h = ops.gemm(a, wp, rowstat=True)
    assert h.rowstats is not None, "an fp16 launch must be able to accumulate row statistics"
    w = _rand((N, C), device, 6, C ** -0.5)
    bias = _rand((N,), device, 7)
    wf, wsum, bf = ops.fold_layernorm(w, gamma, beta, bias, dtype=f16)
    assert wf.dtype == f16
    got = ops.gemm(h, wf, bias=bf, ln=(h.rowstats, wsum, 1e-5))
    assert got is not None and got.dtype == f16
    ref = Fn.layer_norm(h.float(), (C,), gamma, beta, 1e-5) @ w.t() + bias
    _close(got, ref, rtol=6e-3, atol=6e-3, what="f16 folded layernorm")


@pytest.mark.parametrize("heads,d,Sq,Sk,causal,window", [
    (8, 40, 1024, 1024, False, None), (8, 40, 256, 77, False, None), (8, 80, 256, 256, False, None), (8, 160, 64, 77, False, None),
    (8, 40, 12 * 64, 12 * 64, True, (8, 12, 32, 32)), (8, 160, 12 * 16, 12 * 16, True, None),
])
def test_attention_f16(device, heads, d, Sq, Sk, causal, window):
    from seervideoldm_amd import ops
    C = heads * d
    if window is None:
        batch, tq, tk = 3, Sq, Sk
    else:
        ws, F, H, W = window
        batch, tq, tk = 1, F * H * W, F * H * W
    q = _rand((batch * tq, C), device, 1).to(f16)
    k = _rand((batch * tk, C), device, 2).to(f16)
    v = _rand((batch * tk, C), device, 3).to(f16)
    out = torch.empty((batch * tq, C), device=device, dtype=f16)
    ops.attention(q, k, v, out, batch=batch, heads=heads, head_dim=d, Sq=Sq, Sk=Sk, causal=causal, window=window)
    def split(t, n):
        return t.float().reshape(batch, n, heads, d).permute(0, 2, 1, 3)
    if window is None:
        qq, kk, vv = split(q, tq), split(k, tk), split(v, tk)
        s = qq @ kk.transpose(-1, -2) * d ** -0.5
        if causal:
            s = s.masked_fill(torch.ones(Sq, Sk, device=device).triu(1).bool(), float("-inf"))
        ref = (s.softmax(-1) @ vv).permute(0, 2, 1, 3).reshape(batch * tq, C)
    else:
        ws, F, H, W = window
        def win(t):     # [1, F*H*W, C] -> [windows, F*ws*ws, heads, d]
            t = t.float().reshape(F, H // ws, ws, W // ws, ws, heads, d).permute(1, 3, 0, 2, 4, 5, 6)
            return t.reshape(-1, F * ws * ws, heads, d).permute(0, 2, 1, 3)
        qq, kk, vv = win(q), win(k), win(v)
        s = qq @ kk.transpose(-1, -2) * d ** -0.5
        s = s.masked_fill(torch.ones(Sq, Sk, device=device).triu(1).bool(), float("-inf"))
        o = (s.softmax(-1) @ vv).permute(0, 2, 1, 3)                    # [windows, F*ws*ws, heads, d]
        o = o.reshape(H // ws, W // ws, F, ws, ws, heads, d).permute(2, 0, 3, 1, 4, 5, 6)
        ref = o.reshape(F * H * W, C)
    _close(out, ref, rtol=3e-3, atol=2e-3, what=f"f16 attention d{d} Sq{Sq} Sk{Sk}")


def test_small_kernels_f16(device):
    from seervideoldm_amd import ops
    x = _rand((2, 320), device, 1)
    w = _rand((1280, 320), device, 2, 0.05)
    b = _rand((1280,), device, 3)
    y = ops.linear_smallm(x, w.to(f16), b, silu_out=True)
    _close(y, Fn.silu(x @ w.to(f16).float().t() + b), rtol=1e-4, atol=1e-4, what="linear_smallm, fp16 weights")
    c = _rand((77 * 3, 768), device, 4)
    assert torch.equal(ops.cast_bf16(c, f16), c.to(f16)) and torch.equal(ops.cast_bf16(c), c.to(torch.bfloat16))
    # conv behind the nearest-2x upsample as four phase convs
    from seervideoldm_amd.weights import pack_conv3x3_up_phases
    n_img, H, Ci, Co = 3, 8, 128, 128
    xi = _rand((n_img, Ci, H, H), device, 5).to(f16)
    wc = _rand((Co, Ci, 3, 3), device, 6, (9 * Ci) ** -0.5)
    bc = _rand((Co,), device, 7)
    out = ops.conv_up2x(xi.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous(), pack_conv3x3_up_phases(wc).to(f16), n_img, H, H, bias=bc)
    ref = Fn.conv2d(Fn.interpolate(xi.float(), scale_factor=2.0, mode="nearest"), wc, bc, padding=1)
    assert out.dtype == f16
    _close(out, ref.permute(0, 2, 3, 1).reshape(-1, Co), rtol=4e-3, atol=4e-3, what="f16 conv behind nearest-2x")
