"""Per-kernel parity on a real MI355X: every entry point of libseer_hip.so against the fp32 formulas of the
operator it replaces (the same formulas the CPU oracle in oracle/seer_oracle.py uses), on seeded inputs.

Tolerances are for bf16 storage with fp32 accumulation: outputs are rounded once to bf16 (rel 2^-8), inputs are
fed to the reference AFTER rounding to bf16 so only accumulation order and the final rounding differ.
"""
import math

import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _close(got, ref, rtol=2e-2, atol=2e-2, what=""):
    got = got.float()
    ref = ref.float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} outside tol; max err {err.max().item():.4g} "
                             f"(ref max {ref.abs().max().item():.4g}); first bad idx {idx} got {got[tuple(idx)].item():.5g} "
                             f"ref {ref[tuple(idx)].item():.5g}")
    return err.max().item()


# ---------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,tile", [
    (256, 128, 64, 1), (256, 128, 64, 2), (256, 128, 64, 3),
    (384, 320, 320, 0), (1536, 1280, 1280, 0), (1848, 640, 768, 0), (6144, 640, 2560, 0),
    (100, 64, 128, 2), (130, 68, 192, 3), (24576, 320, 320, 1),
    # LDS-direct (global_load_lds) multi-stage variants: tile codes 5..9
    (256, 128, 64, 5), (512, 256, 1280, 5), (512, 256, 1280, 6), (130, 68, 192, 7), (1536, 1280, 1280, 7),
    (100, 64, 128, 8), (384, 320, 2560, 8), (384, 320, 2560, 9), (1848, 640, 768, 9), (6144, 640, 128, 9),
    (384, 320, 2560, 10), (1536, 1280, 1280, 11),      # deeper rings
    (384, 320, 320, 12), (6144, 640, 640, 12), (130, 68, 192, 12), (384, 320, 2560, 13), (1000, 640, 128, 13),   # 160-wide
    (512, 256, 1280, 14), (300, 132, 192, 14), (6144, 640, 640, 14), (1000, 64, 320, 15), (24576, 320, 320, 15),   # 8 waves
    (24576, 320, 1280, 16), (1000, 320, 320, 16), (130, 68, 192, 17), (12288, 640, 640, 18), (100, 64, 128, 18),       # 96-row tiles
    (512, 512, 64, 21), (512, 256, 1280, 21), (300, 132, 192, 21), (6144, 640, 640, 21), (1536, 1280, 128, 21),        # 256x256 ping-pong
])
def test_gemm_plain(device, M, N, K, tile):
    from seervideoldm_amd import ops
    a = _rand((M, K), device, 1).to(bf16)
    w = _rand((N, K), device, 2, K ** -0.5).to(bf16)
    bias = _rand((N,), device, 3)
    res = _rand((M, N), device, 4).to(bf16)
    out = ops.gemm(a, w, bias=bias, residual=res, tile=tile)
    ref = a.float() @ w.float().t() + bias + res.float()
    _close(out, ref, what=f"gemm {M}x{N}x{K} tile{tile}")
    out32 = ops.gemm(a, w, out_f32=True, tile=tile)
    _close(out32, a.float() @ w.float().t(), rtol=2e-3, atol=2e-3, what="gemm f32 out")


@pytest.mark.parametrize("M,N,K,geglu,res,a2k,tile", [
    (24576, 2560, 320, True, False, 0, 0), (24576, 1280, 320, False, False, 0, 0),                 # AUTO routes these to it
    (24576, 960, 320, False, False, 0, 19),                                                        # (AUTO: 160-wide tiles)
    (6144, 5120, 640, True, False, 0, 19), (6144, 1920, 640, False, False, 0, 19), (24576, 320, 320, False, False, 0, 19),
    (24576, 320, 640, False, False, 320, 19), (6144, 640, 640, False, False, 320, 19),            # skip concat (two sources)
    (1000, 320, 320, False, False, 0, 19), (1283, 192, 640, False, False, 0, 19), (2049, 128, 320, True, False, 0, 19),   # row tails
    (4096, 136, 320, False, False, 0, 19), (1536, 200, 640, False, False, 0, 19), (1100, 1000, 320, False, False, 0, 19),  # column tails
    (6144, 640, 640, False, True, 0, 19), (1536, 200, 192, False, False, 0, 19),                    # not eligible: the tile kernel answers
])
def test_gemm_weight_stationary(device, M, N, K, geglu, res, a2k, tile):
    """the weight-stationary persistent kernel (gemm_ws.hip) on the short-K projections of the step and on ragged shapes,
    against fp32; tile 19 asks for it explicitly (shapes it does not take fall through to the tile kernel), 0 checks that AUTO
    routes the wide K = 320 projections to it and gets the same bits"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import geglu_row_order
    K1 = K - a2k
    a = _rand((M, K1), device, 1).to(bf16)
    a2 = _rand((M, a2k), device, 4).to(bf16) if a2k else None
    w = (_rand((N, K), device, 2) * K ** -0.5).to(bf16)
    bias = _rand((N,), device, 3)
    n_out = N // 2 if geglu else N
    r = _rand((M, n_out), device, 5).to(bf16) if res else None
    out = ops.gemm(a, w, a2=a2, bias=bias, residual=r, geglu=geglu, tile=tile, col_scale=None if geglu else (0.5, min(64, N)))
    ws = ops.gemm(a, w, a2=a2, bias=bias, residual=r, geglu=geglu, tile=19, col_scale=None if geglu else (0.5, min(64, N)))
    tiled = ops.gemm(a, w, a2=a2, bias=bias, residual=r, geglu=geglu, tile=20, col_scale=None if geglu else (0.5, min(64, N)))
    if tile == 0:
        assert torch.equal(out, ws), "AUTO must route this shape to the weight-stationary kernel"
    A = a.float() if a2 is None else torch.cat([a.float(), a2.float()], 1)
    acc = A @ w.float().t() + bias
    if geglu:
        # device layout: rows interleaved in groups of 16 (16 value rows, then their 16 gate rows)
        acc = acc.reshape(M, N // 32, 2, 16)
        ref = (acc[:, :, 0] * Fn.gelu(acc[:, :, 1])).reshape(M, N // 2)
    else:
        ref = acc
        ref[:, :min(64, N)] *= 0.5
    if res:
        ref = ref + r.float()
    _close(ws, ref, what=f"ws gemm {M}x{N}x{K}")
    _close(tiled, ref, what=f"tile gemm {M}x{N}x{K}")
    # same K order, same fp32 accumulation: the two kernels agree to the last bf16 rounding
    assert (ws.float() - tiled.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K,splits", [(384, 1280, 1280, 0), (384, 1280, 5120, 4), (200, 68, 1024, 3), (1536, 1280, 5120, 0)])
def test_gemm_split_k(device, M, N, K, splits):
    """split-K (slices -> fp32 workspace -> ordered reduce + epilogue): same answer as one K loop, and deterministic."""
    from seervideoldm_amd import ops
    a = _rand((M, K), device, 1).to(bf16)
    w = _rand((N, K), device, 2, K ** -0.5).to(bf16)
    bias = _rand((N,), device, 3)
    res = _rand((M, N), device, 4).to(bf16)
    rv = _rand((2, N), device, 5)
    ref = a.float() @ w.float().t() + bias + res.float() + rv.repeat_interleave(M // 2, 0)
    o1 = ops.gemm(a, w, bias=bias, residual=res, rowvec=rv, rows_per_batch=M // 2, splits=splits)
    o2 = ops.gemm(a, w, bias=bias, residual=res, rowvec=rv, rows_per_batch=M // 2, splits=splits)
    _close(o1, ref, what=f"split-K gemm {M}x{N}x{K} s{splits}")
    assert torch.equal(o1, o2)
    o3 = ops.gemm(a, w, bias=bias, residual=res, rowvec=rv, rows_per_batch=M // 2, splits=1)
    _close(o3, ref, what="unsplit")


def test_conv3x3_split_k(device):
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    n_img, H, W, Ci, Co = 6, 4, 4, 640, 320
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5).to(bf16)
    bias = _rand((Co,), device, 3)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    ref = Fn.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    for s in (0, 5, 1):
        _close(ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, bias=bias, splits=s), ref, what=f"conv split {s}")


def test_gemm_and_conv_random_shapes_auto_heuristics(device):
    """seeded random shapes through the AUTO tile / split-K heuristics (ragged M, narrow and wide N, short and long K):
    whatever configuration `prepare()` picks must give the same answer as the fp32 formula, deterministically"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    rng = torch.Generator().manual_seed(1234)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
    for it in range(36):
        M = [ri(1, 300), ri(300, 2000), ri(2000, 9000)][it % 3]
        N = 4 * ri(1, 40) if it % 4 else 128 * ri(5, 12)
        K = 64 * [ri(1, 6), ri(6, 40), ri(40, 100)][(it // 3) % 3]
        a = _rand((M, K), device, 10 + it).to(bf16)
        w = _rand((N, K), device, 50 + it, K ** -0.5).to(bf16)
        bias = _rand((N,), device, 90 + it) if it % 2 else None
        res = _rand((M, N), device, 130 + it).to(bf16) if it % 3 == 0 else None
        out = ops.gemm(a, w, bias=bias, residual=res)
        ref = a.float() @ w.float().t() + (bias if bias is not None else 0) + (res.float() if res is not None else 0)
        _close(out, ref, what=f"auto gemm {M}x{N}x{K}")
        assert torch.equal(out, ops.gemm(a, w, bias=bias, residual=res)), f"gemm {M}x{N}x{K} not deterministic"
    for it in range(14):
        n_img, H, W = ri(1, 12), 2 * ri(1, 12), 2 * ri(1, 12)
        Ci, Co = 64 * ri(1, 10), [4 * ri(2, 40), 128 * ri(5, 10)][it % 2]
        stride, up = (2, False) if it % 5 == 0 else ((1, True) if it % 5 == 1 else (1, False))
        x = _rand((n_img, Ci, H, W), device, 200 + it).to(bf16)
        w = _rand((Co, Ci, 3, 3), device, 240 + it, (9 * Ci) ** -0.5).to(bf16)
        bias = _rand((Co,), device, 280 + it)
        x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
        out = ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, stride=stride, upsample=up, bias=bias)
        xin = Fn.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
        ref = Fn.conv2d(xin, w.float(), bias, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
        _close(out, ref, what=f"auto conv n{n_img} {H}x{W} {Ci}->{Co} s{stride} up{up}")
        assert torch.equal(out, ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, stride=stride, upsample=up, bias=bias))


def test_gelu_erf_accuracy(device):
    """the GEGLU epilogue's exact-erf GELU (relu(x) - |x| * 2^(|x| R(|x|) - 1), seer_common.h) against F.gelu over the
    whole useful range: absolute error < 2e-6, and relative error < 2e-3 (half a bf16 ulp) wherever |gelu| > 1e-4"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import interleave_geglu
    C = 64
    gate = torch.linspace(-9, 9, 4096, device=device)
    a = torch.zeros((4096, C), device=device)
    a[:, 0] = 1.0
    a[:, 1] = gate
    w = torch.zeros((2 * 32, C), device=device)      # value rows 0..31, gate rows 32..63
    w[:32, 0] = 1.0                                   # value = 1
    w[32:, 1] = 1.0                                   # gate = a[:, 1]
    wi, bi = interleave_geglu(w.to(bf16), torch.zeros(64, device=device))
    out = ops.gemm(a.to(bf16), wi, bias=bi, geglu=True, out_f32=True)
    g = a.to(bf16)[:, 1].float()
    ref = Fn.gelu(g)[:, None].expand(-1, 32)
    assert (out - ref).abs().max() < 2e-6
    ref64 = (g.double() * 0.5 * (1 + torch.erf(g.double() / math.sqrt(2))))[:, None].expand(-1, 32)
    big = ref64.abs() > 1e-4
    assert ((out.double() - ref64).abs() / ref64.abs().clamp_min(1e-30))[big].max() < 2e-3


def test_gemm_identity_asymmetric(device):
    """A = I against an asymmetric W catches a transposed C write (cdna guide, MFMA section)."""
    from seervideoldm_amd import ops
    K = 128
    a = torch.eye(K, device=device).to(bf16)
    w = (torch.arange(192 * K, device=device).reshape(192, K) % 251).float().to(bf16)   # exact in bf16 (< 256)
    out = ops.gemm(a, w, out_f32=True, tile=2)
    assert torch.equal(out, w.float().t().contiguous())


def test_gemm_dual_source_and_rowvec(device):
    from seervideoldm_amd import ops
    M, K1, K2, N = 768, 640, 320, 320
    a1 = _rand((M, K1), device, 1).to(bf16)
    a2 = _rand((M, K2), device, 2).to(bf16)
    w = _rand((N, K1 + K2), device, 3, 0.03).to(bf16)
    bias = _rand((N,), device, 4)
    rv = _rand((2, N), device, 5)
    out = ops.gemm(a1, w, a2=a2, bias=bias, rowvec=rv, rows_per_batch=M // 2)
    ref = torch.cat([a1, a2], 1).float() @ w.float().t() + bias + rv.repeat_interleave(M // 2, 0)
    _close(out, ref, what="dual-source gemm")


@pytest.mark.parametrize("tile", [0, 21])
@pytest.mark.parametrize("M,C", [(384, 1280), (1536, 320), (2048, 640)])
def test_gemm_geglu(device, M, C, tile):
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import interleave_geglu
    N = 8 * C
    a = _rand((M, C), device, 1).to(bf16)
    w = _rand((N, C), device, 2, C ** -0.5).to(bf16)
    bias = _rand((N,), device, 3, 0.5)
    wi, bi = interleave_geglu(w, bias)
    out = ops.gemm(a, wi, bias=bi, geglu=True, tile=tile)
    h = a.float() @ w.float().t() + bias
    val, gate = h.chunk(2, dim=-1)
    ref = val * Fn.gelu(gate)
    _close(out, ref, what="geglu")


def test_gemm_strided_views(device):
    """A and C as column slices of wider buffers (fused qkv layout)."""
    from seervideoldm_amd import ops
    M, K, N = 512, 320, 320
    big = _rand((M, 3 * K), device, 1).to(bf16)
    w = _rand((N, K), device, 2, 0.05).to(bf16)
    outbig = torch.zeros((M, 2 * N), device=device, dtype=bf16)
    ops.gemm(big[:, K:2 * K], w, out=outbig[:, N:])
    _close(outbig[:, N:], big[:, K:2 * K].float() @ w.float().t(), what="strided gemm")
    assert (outbig[:, :N] == 0).all()


def test_gemm_batched_and_transposed(device):
    from seervideoldm_amd import ops
    Bt, M, N, K = 3, 256, 192, 128
    a = _rand((Bt, M, K), device, 1).to(bf16)
    w = _rand((Bt, N, K), device, 2, 0.1).to(bf16)
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float())
    _close(ops.gemm_batched(a, w), ref, what="batched")
    _close(ops.gemm_batched(a, w, trans_out=True), ref.transpose(1, 2), what="batched trans")


# ---------------------------------------------------------------------------------------------------- conv
@pytest.mark.parametrize("n_img,H,W,Ci,Co,stride,up", [
    (2, 8, 8, 64, 64, 1, False), (3, 16, 16, 128, 64, 1, False), (2, 16, 16, 64, 128, 2, False),
    (2, 8, 8, 64, 64, 1, True), (24, 32, 32, 320, 320, 1, False), (4, 4, 4, 1280, 1280, 1, False),
    (2, 6, 10, 64, 68, 1, False),
])
@pytest.mark.parametrize("tile", [0, 7, 9, 12, 13, 14, 15, 16, 17, 18, 21])
def test_conv3x3(device, n_img, H, W, Ci, Co, stride, up, tile):
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5).to(bf16)
    bias = _rand((Co,), device, 3)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    out = ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, stride=stride, upsample=up, bias=bias, tile=tile, splits=1)
    xin = x.float()
    if up:
        xin = Fn.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = Fn.conv2d(xin, w.float(), bias, stride=stride, padding=1)
    ref_cl = ref.permute(0, 2, 3, 1).reshape(-1, Co)
    _close(out, ref_cl, what=f"conv {Ci}->{Co} s{stride} up{up}")


@pytest.mark.parametrize("n_img,H,W,Ci,Co", [(2, 8, 8, 64, 64), (3, 4, 4, 128, 192), (24, 16, 16, 640, 640), (2, 6, 10, 64, 68),
                                             (24, 4, 4, 1280, 1280)])
@pytest.mark.parametrize("tile", [0, 5, 7, 8, 12, 14, 21])
def test_conv_up2x_phases(device, n_img, H, W, Ci, Co, tile):
    """nearest-2x + conv3x3 (Upsample3D, resnet.py:52-57) as four 2x2 phase convs against F.interpolate + F.conv2d in fp32,
    and against the 9-tap kernel on the same input (the two differ by the rounding of the summed taps only)"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3, pack_conv3x3_up_phases
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5)
    bias = _rand((Co,), device, 3)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    out = ops.conv_up2x(x_cl, pack_conv3x3_up_phases(w).to(bf16), n_img, H, W, bias=bias, tile=tile)
    ref = Fn.conv2d(Fn.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.to(bf16).float(), bias, padding=1)
    ref_cl = ref.permute(0, 2, 3, 1).reshape(-1, Co)
    _close(out, ref_cl, what=f"conv_up2x {Ci}->{Co} {H}x{W} tile{tile}")
    nine = ops.conv3x3(x_cl, pack_conv3x3(w.to(bf16)), n_img, H, W, upsample=True, bias=bias, splits=1)
    assert (out.float() - nine.float()).abs().max().item() <= 3e-2 * ref_cl.abs().max().item()


@pytest.mark.parametrize("tile", [0, 2, 7, 8])
def test_conv3x3_pad_after_only(device, tile):
    """the VAE encoder's Downsample: F.pad(x, (0,1,0,1)) + conv(stride 2, padding 0) (ldm .../model.py:60-78)"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    n_img, H, W, Ci, Co = 3, 16, 24, 128, 64
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5).to(bf16)
    bias = _rand((Co,), device, 3)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    out = ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, stride=2, pad_after_only=True, bias=bias, tile=tile, splits=1)
    ref = Fn.conv2d(Fn.pad(x.float(), (0, 1, 0, 1)), w.float(), bias, stride=2, padding=0)
    assert out.shape[0] == n_img * (H // 2) * (W // 2)
    _close(out, ref.permute(0, 2, 3, 1).reshape(-1, Co), what="conv s2 padded after only")


def test_conv3x3_epilogue(device):
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    B, Fr, H, W, Ci, Co = 2, 3, 8, 8, 64, 128
    n_img = B * Fr
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, 0.04).to(bf16)
    bias = _rand((Co,), device, 3)
    temb = _rand((B, Co), device, 4)
    res = _rand((n_img * H * W, Co), device, 5).to(bf16)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    out = ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, bias=bias, rowvec=temb, rows_per_batch=Fr * H * W,
                      residual=res)
    ref = Fn.conv2d(x.float(), w.float(), bias, padding=1) + temb.repeat_interleave(Fr, 0)[:, :, None, None]
    ref_cl = ref.permute(0, 2, 3, 1).reshape(-1, Co) + res.float()
    _close(out, ref_cl, what="conv epilogue")


# ---------------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, causal):
    d = q.shape[-1]
    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), k.float()) * d ** -0.5
    if causal:
        m = torch.ones(s.shape[-2:], dtype=torch.bool, device=s.device).tril()
        s = s.masked_fill(~m, float("-inf"))
    return torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), v.float())


@pytest.mark.parametrize("d,Sq,Sk,causal", [
    (40, 128, 128, False), (40, 1024, 1024, False), (80, 256, 256, False), (160, 64, 64, False), (160, 16, 16, False),
    (40, 1024, 77, False), (80, 256, 77, False), (160, 64, 77, False), (160, 16, 77, False),
    (40, 768, 768, True), (80, 192, 192, True), (160, 192, 192, True), (40, 100, 100, True), (80, 272, 272, True),
    (96, 77, 77, False), (96, 924, 77, False), (96, 12, 12, True), (96, 200, 200, True),     # FSTextTransformer head dim
])
def test_attention(device, d, Sq, Sk, causal):
    from seervideoldm_amd import ops
    B, Hh = 3, 8
    C = Hh * d
    q = _rand((B, Sq, Hh, d), device, 1).to(bf16)
    k = _rand((B, Sk, Hh, d), device, 2).to(bf16)
    v = _rand((B, Sk, Hh, d), device, 3).to(bf16)
    out = torch.zeros((B * Sq, C), device=device, dtype=bf16)
    ops.attention(q.reshape(B * Sq, C), k.reshape(B * Sk, C), v.reshape(B * Sk, C), out, batch=B, heads=Hh,
                  head_dim=d, Sq=Sq, Sk=Sk, causal=causal)
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), causal)
    ref = ref.permute(0, 2, 1, 3).reshape(B * Sq, C)
    _close(out, ref, rtol=2e-2, atol=1e-2, what=f"attn d{d} {Sq}x{Sk} causal={causal}")


def test_attention_fused_qkv_layout(device):
    """q, k, v as column slices of one [tokens, 3C] buffer (what the fused projection GEMM writes)."""
    from seervideoldm_amd import ops
    B, S, Hh, d = 2, 256, 8, 80
    C = Hh * d
    qkv = _rand((B * S, 3 * C), device, 7).to(bf16)
    out = torch.empty((B * S, C), device=device, dtype=bf16)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, batch=B, heads=Hh, head_dim=d, Sq=S, Sk=S)
    q, k, v = [t.reshape(B, S, Hh, d).permute(0, 2, 1, 3) for t in qkv.split(C, dim=1)]
    ref = _attn_ref(q, k, v, False).permute(0, 2, 1, 3).reshape(B * S, C)
    _close(out, ref, rtol=2e-2, atol=1e-2, what="fused-qkv attention")


@pytest.mark.parametrize("d,S,window", [(40, 1024, None), (40, 300, None), (80, 256, None), (40, 3 * 64, (8, 3, 16, 16))])
def test_attention_head_major_operands(device, d, S, window):
    """seer_attn_desc::q_hs / k_hs / v_hs: the same q, k, v handed over HEAD-MAJOR ([batch][head][tokens][d] contiguous) give the
    bits of the token-major call -- same arithmetic, other addresses -- in both kernels, plain and windowed-causal"""
    from seervideoldm_amd import ops
    B, Hh = 2, 8
    C = Hh * d
    tok = S if window is None else window[1] * window[2] * window[3]
    qkv = _rand((B * tok, 3 * C), device, 17).to(bf16)
    kw = dict(batch=B, heads=Hh, head_dim=d, Sq=S, Sk=S, causal=window is not None, window=window)
    want = torch.empty((B * tok, C), device=device, dtype=bf16)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], want, **kw)
    hm = [t.reshape(B, tok, Hh, d).permute(0, 2, 1, 3).contiguous().reshape(B * Hh * tok, d) for t in qkv.split(C, dim=1)]
    got = torch.empty_like(want)
    ops.attention(qkv[:, :C], hm[1], hm[2], got, kv_head_major=True, **kw)
    assert torch.equal(got, want)
    got2 = torch.empty_like(want)
    ops.attention(hm[0], hm[1], hm[2], got2, q_head_major=True, kv_head_major=True, **kw)
    assert torch.equal(got2, want)


def test_attention_random_shapes(device):
    """seeded ragged sequence lengths (tails in the last query block and the last key tile), causal with a query offset"""
    from seervideoldm_amd import ops
    rng = torch.Generator().manual_seed(77)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
    for it in range(16):
        d = (40, 80, 96, 160)[it % 4]
        B, Hh = ri(1, 3), ri(1, 8)
        Sq, Sk = ri(1, 500), ri(1, 500)
        causal = it % 3 == 0
        off = 0
        if causal:
            Sk = max(Sk, Sq)
            off = ri(0, Sk - Sq)
        C = Hh * d
        q = _rand((B, Sq, Hh, d), device, 300 + it).to(bf16)
        k = _rand((B, Sk, Hh, d), device, 340 + it).to(bf16)
        v = _rand((B, Sk, Hh, d), device, 380 + it).to(bf16)
        out = torch.zeros((B * Sq, C), device=device, dtype=bf16)
        ops.attention(q.reshape(B * Sq, C), k.reshape(B * Sk, C), v.reshape(B * Sk, C), out, batch=B, heads=Hh, head_dim=d,
                      Sq=Sq, Sk=Sk, causal=causal, causal_offset=off)
        qq, kk, vv = [t.permute(0, 2, 1, 3).float() for t in (q, k, v)]
        s = torch.einsum("bhqd,bhkd->bhqk", qq, kk) * d ** -0.5
        if causal:
            i = torch.arange(Sq, device=device)[:, None] + off
            j = torch.arange(Sk, device=device)[None, :]
            s = s.masked_fill(~(j <= i), float("-inf"))
        ref = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), vv).permute(0, 2, 1, 3).reshape(B * Sq, C)
        _close(out, ref, rtol=2e-2, atol=1e-2, what=f"attn d{d} B{B} H{Hh} {Sq}x{Sk} causal={causal}+{off}")


def test_attention_strided_sequences(device):
    """rows ordered (frame, token): causal attention over the frames of every token position, read through strides
    (seq stride = tokens per frame, batch stride = one row) -- FSTextTransformer's temporal block"""
    from seervideoldm_amd import ops
    Fr, L, Hh, d = 12, 77, 8, 96
    C = Hh * d
    qkv = _rand((Fr * L, 3 * C), device, 5).to(bf16)
    out = torch.zeros((Fr * L, C), device=device, dtype=bf16)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, batch=L, heads=Hh, head_dim=d, Sq=Fr, Sk=Fr,
                  causal=True, seq_stride_rows=L, batch_stride_rows=1)
    q, k, v = [t.reshape(Fr, L, Hh, d).permute(1, 2, 0, 3) for t in qkv.split(C, dim=1)]     # [L, heads, F, d]
    ref = _attn_ref(q, k, v, True).permute(2, 0, 1, 3).reshape(Fr * L, C)
    _close(out, ref, rtol=2e-2, atol=1e-2, what="strided-sequence attention")


def test_attention_softmax_spike(device):
    """force a late running-max jump (online-softmax rescale path) with a spiked key."""
    from seervideoldm_amd import ops
    B, S, Hh, d = 1, 256, 8, 40
    C = Hh * d
    q = _rand((B, S, Hh, d), device, 1).to(bf16)
    k = _rand((B, S, Hh, d), device, 2).to(bf16)
    v = _rand((B, S, Hh, d), device, 3).to(bf16)
    k[:, 200] = (q[:, 5] * 4).to(bf16)        # key 200 dominates query 5 only in a late tile
    out = torch.empty((B * S, C), device=device, dtype=bf16)
    ops.attention(q.reshape(S, C), k.reshape(S, C), v.reshape(S, C), out, batch=B, heads=Hh, head_dim=d, Sq=S, Sk=S)
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), False)
    _close(out, ref.permute(0, 2, 1, 3).reshape(S, C), rtol=2e-2, atol=1e-2, what="spiked softmax")


def _rel_l2(got, ref):
    got, ref = got.float(), ref.float()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("variant", [1, 2, 3, 5, 7])
@pytest.mark.parametrize("Sq,Sk,causal", [(1024, 1024, False), (768, 768, True), (1000, 930, False), (333, 333, True),
                                           (130, 2049, False), (64, 256, False), (256, 128, False), (512, 1152, False),
                                           (4096, 4096, False)])      # BASELINE config 4: spatial attention of a 64x64 latent
def test_attention_d40_kernels(device, variant, Sq, Sk, causal):
    """head_dim 40: the generic kernel (variant 1), the d = 40 kernel's fast path (3; 2 = its 64-queries-per-wave shape; 7 = that shape
    on the three-stage K|V ring, whole tiles only: one, nine and many tiles) and its tracked form (5) against the fp32 formula
    (xformers MEA as called at attention.py:622-630)"""
    from seervideoldm_amd import ops
    from seervideoldm_amd._lib import SeerHipError
    B, Hh, d = 2, 8, 40
    C = Hh * d
    if variant == 7 and (causal or Sq % 256 or Sk % 128):
        x = torch.zeros((B * max(Sq, Sk), C), device=device, dtype=bf16)
        with pytest.raises(SeerHipError):        # the ring form refuses what it cannot run instead of running something else
            ops.attention(x[:B * Sq], x[:B * Sk], x[:B * Sk], torch.zeros((B * Sq, C), device=device, dtype=bf16), batch=B, heads=Hh,
                          head_dim=d, Sq=Sq, Sk=Sk, causal=causal, causal_offset=(Sk - Sq if causal else 0), variant=7)
        return
    q = _rand((B, Sq, Hh, d), device, 1).to(bf16)
    k = _rand((B, Sk, Hh, d), device, 2).to(bf16)
    v = _rand((B, Sk, Hh, d), device, 3).to(bf16)
    out = torch.zeros((B * Sq, C), device=device, dtype=bf16)
    ops.attention(q.reshape(B * Sq, C), k.reshape(B * Sk, C), v.reshape(B * Sk, C), out, batch=B, heads=Hh,
                  head_dim=d, Sq=Sq, Sk=Sk, causal=causal, causal_offset=(Sk - Sq if causal else 0), variant=variant)
    qq, kk, vv = [t.permute(0, 2, 1, 3).float() for t in (q, k, v)]
    s = torch.einsum("bhqd,bhkd->bhqk", qq, kk) * d ** -0.5
    if causal:
        i = torch.arange(Sq, device=device)[:, None] + (Sk - Sq)
        s = s.masked_fill(~(torch.arange(Sk, device=device)[None, :] <= i), float("-inf"))
    ref = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), vv).permute(0, 2, 1, 3).reshape(B * Sq, C)
    _close(out, ref, rtol=2e-2, atol=1e-2, what=f"d40 variant {variant} {Sq}x{Sk} causal={causal}")


@pytest.mark.parametrize("d,variant", [(40, 1), (40, 2), (40, 3), (40, 5), (40, 7), (80, 0), (96, 0), (160, 0)])
@pytest.mark.parametrize("amp", [3.0, 6.0])
def test_attention_sharp_softmax(device, d, variant, amp):
    """scores hundreds of log2 units apart (cdna guide, rule 26: the rare branch needs its own test).  This is the case that
    exposed the dropped half of the running maximum (the second result of the permlane32_swap builtin) as NaN rows, and it
    drives the d = 40 fast path into its overflow fallback."""
    from seervideoldm_amd import ops
    B, S, Hh = 2, 1024, 8
    C = Hh * d
    q = _rand((B, S, Hh, d), device, 31, amp).to(bf16)
    k = _rand((B, S, Hh, d), device, 32, amp).to(bf16)
    v = _rand((B, S, Hh, d), device, 33, amp).to(bf16)
    out = torch.zeros((B * S, C), device=device, dtype=bf16)
    ops.attention(q.reshape(B * S, C), k.reshape(B * S, C), v.reshape(B * S, C), out, batch=B, heads=Hh, head_dim=d,
                  Sq=S, Sk=S, variant=variant)
    assert torch.isfinite(out.float()).all(), "non-finite attention output"
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), False)
    ref = ref.permute(0, 2, 1, 3).reshape(B * S, C)
    # scores of magnitude ~amp^2 * sqrt(d): one ulp of a bf16 operand moves a score by ~2^-9 * |s|, and the softmax is
    # nearly one-hot, so single rows can flip between two keys; the L2 norm is the meaningful measure.  The d = 40 kernel
    # multiplies q by scale * log2(e) before the MFMA (one more rounding of q when the caller did not prescale it).
    assert _rel_l2(out, ref) < (3e-2 if (d == 40 and variant != 1) else 1e-2)


@pytest.mark.parametrize("d,S", [(40, 1024), (40, 77), (80, 256), (160, 64)])
def test_attention_q_prescaled(device, d, S):
    """q handed over as q * scale * log2(e) (SEER_ATTN_Q_PRESCALED): the kernels exponentiate the raw dot products"""
    from seervideoldm_amd import ops
    B, Hh = 3, 8
    C = Hh * d
    q = (_rand((B, S, Hh, d), device, 41) * ops.qk_prescale(d)).to(bf16)
    k = _rand((B, S, Hh, d), device, 42).to(bf16)
    v = _rand((B, S, Hh, d), device, 43).to(bf16)
    out = torch.zeros((B * S, C), device=device, dtype=bf16)
    ops.attention(q.reshape(B * S, C), k.reshape(B * S, C), v.reshape(B * S, C), out, batch=B, heads=Hh, head_dim=d,
                  Sq=S, Sk=S, q_prescaled=True)
    s = torch.einsum("bhqd,bhkd->bhqk", q.permute(0, 2, 1, 3).float(), k.permute(0, 2, 1, 3).float()) * math.log(2.0)
    ref = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), v.permute(0, 2, 1, 3).float()).permute(0, 2, 1, 3).reshape(B * S, C)
    _close(out, ref, rtol=2e-2, atol=1e-2, what=f"prescaled q d{d}")


def test_attention_d40_lse(device):
    """the log-sum-exp the training step asks for: on an un-prescaled q it comes from the kernel that scales the fp32 scores (the
    arithmetic seer_attn_bwd rebuilds P with); with variant 5 from the tracked form of the d = 40 kernel.  Both are checked, and
    exp2(q.k * scale * log2(e) - lse) -- the backward's P -- must sum to 1 over the keys with the default routing."""
    from seervideoldm_amd import ops
    B, S, Hh, d = 2, 512, 8, 40
    C = Hh * d
    q = _rand((B, S, Hh, d), device, 51).to(bf16)
    k = _rand((B, S, Hh, d), device, 52).to(bf16)
    v = _rand((B, S, Hh, d), device, 53).to(bf16)
    out = torch.zeros((B * S, C), device=device, dtype=bf16)
    lse = torch.zeros((B * Hh, S), device=device, dtype=torch.float32)
    ops.attention(q.reshape(B * S, C), k.reshape(B * S, C), v.reshape(B * S, C), out, batch=B, heads=Hh, head_dim=d,
                  Sq=S, Sk=S, lse=lse)
    s = torch.einsum("bhqd,bhkd->bhqk", q.permute(0, 2, 1, 3).float(), k.permute(0, 2, 1, 3).float()) * d ** -0.5
    ref = torch.logsumexp(s, -1) / math.log(2.0)            # log2 domain
    assert (lse.reshape(B, Hh, S) - ref).abs().max().item() < 2e-2
    psum = torch.exp2(s / math.log(2.0) - lse.reshape(B, Hh, S, 1)).sum(-1)
    assert (psum - 1).abs().max().item() < 2e-3, "the backward's probabilities must sum to 1"
    lse5 = torch.zeros_like(lse)
    ops.attention(q.reshape(B * S, C), k.reshape(B * S, C), v.reshape(B * S, C), out, batch=B, heads=Hh, head_dim=d,
                  Sq=S, Sk=S, lse=lse5, variant=5)
    assert (lse5.reshape(B, Hh, S) - ref).abs().max().item() < 2e-2


def test_gemm_col_scale(device):
    """SEER_EPI_COLSCALE: the first columns of a projection scaled in the epilogue, also through split-K"""
    from seervideoldm_amd import ops
    for M, N, K, cols, splits in ((512, 960, 320, 320, 1), (384, 1280, 2560, 1280, 4), (200, 384, 128, 128, 1)):
        a = _rand((M, K), device, 61).to(bf16)
        w = (_rand((N, K), device, 62) * K ** -0.5).to(bf16)
        out = ops.gemm(a, w, col_scale=(0.228, cols), splits=splits)
        ref = a.float() @ w.float().t()
        ref[:, :cols] *= 0.228
        _close(out, ref, what=f"col_scale {M}x{N}x{K}")


@pytest.mark.parametrize("d,Fr,H,W,ws", [(40, 4, 32, 32, 8), (80, 12, 16, 16, 4), (160, 3, 8, 8, 4),
                                         (80, 12, 32, 32, 8)])   # config 4's second level: 768 causal keys per window at d = 80
def test_window_attention(device, d, Fr, H, W, ws):
    """temporal window attention == window_partition -> causal attention -> window_reverse (attention.py:42-69,661-703)."""
    from seervideoldm_amd import ops
    B, Hh = 2, 8
    C = Hh * d
    T = Fr * H * W
    qkv = _rand((B * T, 3 * C), device, 11).to(bf16)
    out = torch.zeros((B * T, C), device=device, dtype=bf16)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, batch=B, heads=Hh, head_dim=d,
                  Sq=Fr * ws * ws, Sk=Fr * ws * ws, causal=True, window=(ws, Fr, H, W))

    def part(t):   # [B*T, C] -> [nW*B, heads, F*ws*ws, d]
        t = t.float().reshape(B, Fr, H // ws, ws, W // ws, ws, Hh, d)
        t = t.permute(2, 4, 0, 6, 1, 3, 5, 7)      # nwy nwx B heads F wy wx d
        return t.reshape(-1, Hh, Fr * ws * ws, d)
    q, k, v = [part(t) for t in qkv.split(C, dim=1)]
    o = _attn_ref(q, k, v, True)                   # [nW*B, heads, S, d]
    o = o.reshape(H // ws, W // ws, B, Hh, Fr, ws, ws, d).permute(2, 4, 0, 5, 1, 6, 3, 7).reshape(B * T, C)
    _close(out, o, rtol=2e-2, atol=1e-2, what="window attention")


@pytest.mark.parametrize("d,Fr,H,W,ws,f0,f1", [(40, 6, 16, 16, 4, 2, 4), (80, 5, 8, 8, 4, 3, 5), (160, 4, 4, 4, 0, 1, 3),
                                               (40, 4, 32, 32, 8, 0, 2)])
def test_frame_shard_attention(device, d, Fr, H, W, ws, f0, f1):
    """frame-sharded temporal attention: queries of frames [f0, f1) against the K|V of all frames with
    causal_offset = position of frame f0 must equal the rows [f0, f1) of the unsharded causal attention."""
    from seervideoldm_amd import ops
    B, Hh = 2, 8
    C = Hh * d
    T = Fr * H * W
    qkv = _rand((B * T, 3 * C), device, 21).to(bf16)
    full = torch.zeros((B * T, C), device=device, dtype=bf16)
    if ws:
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], full, batch=B, heads=Hh, head_dim=d,
                      Sq=Fr * ws * ws, Sk=Fr * ws * ws, causal=True, window=(ws, Fr, H, W))
    else:
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], full, batch=B, heads=Hh, head_dim=d,
                      Sq=T, Sk=T, causal=True)
    Fl = f1 - f0
    ql = qkv.reshape(B, Fr, H * W, 3 * C)[:, f0:f1, :, :C].reshape(B * Fl * H * W, C).contiguous()
    out = torch.zeros((B * Fl * H * W, C), device=device, dtype=bf16)
    if ws:
        ops.attention(ql, qkv[:, C:2 * C], qkv[:, 2 * C:], out, batch=B, heads=Hh, head_dim=d, Sq=Fl * ws * ws,
                      Sk=Fr * ws * ws, causal=True, window=(ws, Fr, H, W), Fq=Fl, causal_offset=f0 * ws * ws)
    else:
        ops.attention(ql, qkv[:, C:2 * C], qkv[:, 2 * C:], out, batch=B, heads=Hh, head_dim=d, Sq=Fl * H * W, Sk=T,
                      causal=True, causal_offset=f0 * H * W)
    ref = full.reshape(B, Fr, H * W, C)[:, f0:f1].reshape(B * Fl * H * W, C)
    assert torch.equal(out, ref), "a frame shard must reproduce the unsharded rows bit for bit"


@pytest.mark.parametrize("d,T,off", [(40, 640, 0), (80, 192, 64), (160, 48, 16)])
def test_gemm_rotary_epilogue(device, d, T, off):
    """q|k|v projection with rotary fused into the epilogue == projection followed by the rotary kernel"""
    from seervideoldm_amd import ops
    B, Hh = 2, 8
    C = Hh * d
    x = _rand((B * T, C), device, 1).to(bf16)
    w = _rand((3 * C, C), device, 2, C ** -0.5).to(bf16)
    freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(device)
    cs = ops.rotary_table(freqs, T + off)
    fused = ops.gemm(x, w, rotary=(cs, T, off, d, 32, 2 * C))
    plain32 = ops.gemm(x, w, out_f32=True)
    # reference: rotate the fp32 projection, round once (what the fused epilogue does)
    pos = (torch.arange(B * T, device=device) % T) + off
    c, s = cs[pos, :, 0], cs[pos, :, 1]
    t = plain32[:, :2 * C].reshape(B * T, 2 * Hh, d)
    x0, x1 = t[..., :32:2], t[..., 1:32:2]
    rot = torch.stack([x0 * c[:, None] - x1 * s[:, None], x1 * c[:, None] + x0 * s[:, None]], -1).flatten(-2)
    ref = torch.cat([torch.cat([rot, t[..., 32:]], -1).reshape(B * T, 2 * C), plain32[:, 2 * C:]], 1)
    _close(fused, ref, rtol=1e-2, atol=1e-2, what="fused rotary")


def test_rotary(device):
    from seervideoldm_amd import ops
    B, T, Hh, d = 2, 640, 8, 40
    C = Hh * d
    x = _rand((B * T, 3 * C), device, 1).to(bf16)
    freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(device)
    cs = ops.rotary_table(freqs, T)
    y = x.clone()
    ops.rotary_inplace(y, 0, C, Hh, d, 32, T, cs)
    pos = torch.arange(T, device=device).float()
    ang = torch.einsum("t,f->tf", pos, freqs).repeat_interleave(2, dim=-1)       # [T, 32]
    def rot(t):    # [B*T, C]
        t = t.float().reshape(B, T, Hh, d)
        tr = t[..., :32]
        x1, x2 = tr[..., 0::2], tr[..., 1::2]
        rh = torch.stack((-x2, x1), dim=-1).flatten(-2)
        tr = tr * ang.cos()[None, :, None, :] + rh * ang.sin()[None, :, None, :]
        return torch.cat([tr, t[..., 32:]], -1).reshape(B * T, C)
    _close(y[:, :C], rot(x[:, :C]), what="rotary q")
    _close(y[:, C:2 * C], rot(x[:, C:2 * C]), what="rotary k")
    assert torch.equal(y[:, 2 * C:], x[:, 2 * C:])


# ---------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,rows,C1,C2,silu", [(2, 768, 320, 0, True), (2, 200, 1280, 640, True), (1, 64, 2560, 0, False),
                                               (3, 1000, 128, 0, True), (2, 192, 640, 320, False)])
def test_groupnorm(device, B, rows, C1, C2, silu):
    from seervideoldm_amd import ops
    x1 = (_rand((B * rows, C1), device, 1) * 2 + 0.5).to(bf16)
    x2 = (_rand((B * rows, C2), device, 2) - 1.0).to(bf16) if C2 else None
    Ct = C1 + C2
    gamma = _rand((Ct,), device, 3) + 1.0
    beta = _rand((Ct,), device, 4)
    stats = torch.zeros((B, 32, 2), device=device, dtype=torch.float32)
    ops.groupnorm_stats(x1, x2, B, 32, stats)
    y = ops.groupnorm_apply(x1, x2, B, 32, stats, rows * (Ct // 32), 1e-5, gamma, beta, silu)
    xc = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    xr = xc.reshape(B, rows, Ct).permute(0, 2, 1)            # [B, C, rows]
    ref = Fn.group_norm(xr, 32, gamma, beta, eps=1e-5)
    if silu:
        ref = Fn.silu(ref)
    _close(y, ref.permute(0, 2, 1).reshape(B * rows, Ct), what="groupnorm")


def _colsum_reference(y, B, groups):
    """(sum, sumsq) per (b, group) of the bf16 tensor y [B*rows, C] in float64."""
    rows, Ct = y.shape[0] // B, y.shape[1]
    v = y.double().reshape(B, rows, groups, Ct // groups)
    return torch.stack([v.sum(dim=(1, 3)), (v * v).sum(dim=(1, 3))], -1)


@pytest.mark.parametrize("kind,B,shape,tile", [
    ("conv", 2, (4, 32, 32, 320, 320, 1), 0),          # 96x160 / 128x160 tile, bias + row vector + residual
    ("conv", 2, (6, 16, 16, 640, 640, 1), 5),          # 128x128 tile, long K (AUTO would split K: no column sums there)
    ("conv", 2, (4, 8, 8, 1280, 1280, 1), 8),          # 64x64 ring, 128 rows
    ("conv", 2, (4, 32, 32, 320, 320, 2), 8),          # stride 2 (downsampler), 64x64 ring
    ("conv", 3, (3, 16, 16, 320, 640, 1), 5),          # 128x128 tile, batch of 3
    ("conv", 2, (4, 16, 16, 320, 320, 1), 2),          # register-staged 64x64
    ("gemm", 2, (4096, 320, 320), 0),                  # proj_out + residual, short K
    ("gemm", 2, (2048, 640, 640), 0),
    ("gemm", 2, (1024, 1280, 1280), 0),
    ("gemm", 2, (6144, 640, 640), 7),                  # 128x64 ring
    ("up", 2, (4, 8, 8, 640, 640), 0),                 # four phase convs
    ("up", 2, (2, 16, 16, 320, 320), 0),
    ("conv", 2, (24, 8, 8, 1280, 1280, 1), 0),         # AUTO: split-K (128x128 x 4 slices), sums from the reduce pass, 16-row partials
    ("conv", 2, (24, 4, 4, 1280, 1280, 1), 0),         # AUTO: split-K on 384 rows, 4-row partials
    ("conv", 2, (6, 16, 16, 640, 640, 1), 0),          # AUTO: split-K at the 16x16 level of a frame shard
    ("gemm", 2, (384, 1280, 5120), 0),                 # AUTO: split-K plain GEMM (deep-level feed-forward output)
])
def test_groupnorm_stats_from_colsums(device, kind, B, shape, tile):
    """The column sums a GEMM / conv leaves next to its output give the same GroupNorm statistics as the pass over the output
    (seer_groupnorm_stats), and both match float64 sums of the stored bf16 values."""
    from seervideoldm_amd import ops
    G = 32
    if kind == "conv":
        n_img, H, W, Ci, Co, stride = shape
        x = _rand((n_img * H * W, Ci), device, 1).to(bf16)
        w = (_rand((Co, 9 * Ci), device, 2) / math.sqrt(9 * Ci)).to(bf16)
        Ho = (H - 1) // stride + 1
        rows_pb = n_img // B * Ho * Ho if n_img % B == 0 else None
        if rows_pb is None:
            pytest.skip("images do not split over the batch")
        res = _rand((n_img * Ho * Ho, Co), device, 3).to(bf16) if stride == 1 else None
        temb = _rand((B, Co), device, 4)
        y = ops.conv3x3(x, w, n_img, H, W, stride=stride, bias=_rand((Co,), device, 5), residual=res, rowvec=temb,
                        rows_per_batch=rows_pb, tile=tile, splits=1 if tile else 0, colsum_batch=B)
    elif kind == "gemm":
        M, N, K = shape
        a = _rand((M, K), device, 1).to(bf16)
        w = (_rand((N, K), device, 2) / math.sqrt(K)).to(bf16)
        y = ops.gemm(a, w, bias=_rand((N,), device, 5), residual=_rand((M, N), device, 3).to(bf16), tile=tile, colsum_batch=B)
    else:
        from seervideoldm_amd.weights import pack_conv3x3_up_phases
        n_img, H, W, Ci, Co = shape
        x = _rand((n_img * H * W, Ci), device, 1).to(bf16)
        w = _rand((Co, Ci, 3, 3), device, 2) / math.sqrt(9 * Ci)
        y = ops.conv_up2x(x, pack_conv3x3_up_phases(w).to(bf16), n_img, H, W, bias=_rand((Co,), device, 5), tile=tile,
                          colsum_batch=B)
    cs = y.colsums
    assert cs is not None, "this launch was expected to produce column sums"
    got = torch.zeros((B, G, 2), device=device, dtype=torch.float32)
    ops.groupnorm_stats_from_colsums(cs, None, B, G, got)
    two_stage = torch.zeros_like(got)
    ops.groupnorm_stats(y, None, B, G, two_stage)
    ref = _colsum_reference(y, B, G)
    scale = ref[..., 1].abs().max().item() + 1.0
    assert (got.double() - ref).abs().max().item() <= 2e-5 * scale, (got.double() - ref).abs().max().item()
    assert (got - two_stage).abs().max().item() <= 4e-5 * scale
    # deterministic: a second launch leaves bit-identical sums
    first = cs.buf.clone()
    if kind == "gemm":
        y2 = ops.gemm(a, w, bias=_rand((N,), device, 5), residual=_rand((M, N), device, 3).to(bf16), tile=tile, colsum_batch=B)
        assert torch.equal(y2.colsums.buf, first) and torch.equal(y2, y)


def test_groupnorm_stats_from_colsums_concat(device):
    """Two sources with different producers (a conv on 64x64 tiles and a GEMM on 96-row tiles) and a group that straddles the
    concat boundary: 640 + 320 channels in 32 groups of 30."""
    from seervideoldm_amd import ops
    B, n_img, H = 2, 4, 16
    xa = _rand((n_img * H * H, 320), device, 1).to(bf16)
    wa = (_rand((640, 9 * 320), device, 2) / math.sqrt(9 * 320)).to(bf16)
    ya = ops.conv3x3(xa, wa, n_img, H, H, bias=_rand((640,), device, 3), tile=8, splits=1, colsum_batch=B)   # (AUTO would split K)
    a = _rand((n_img * H * H, 320), device, 4).to(bf16)
    wb = (_rand((320, 320), device, 5) / math.sqrt(320)).to(bf16)
    yb = ops.gemm(a, wb, bias=_rand((320,), device, 6), residual=_rand((n_img * H * H, 320), device, 7).to(bf16), colsum_batch=B)
    assert ya.colsums is not None and yb.colsums is not None
    got = torch.zeros((B, 32, 2), device=device, dtype=torch.float32)
    ops.groupnorm_stats_from_colsums(ya.colsums, yb.colsums, B, 32, got)
    ref = _colsum_reference(torch.cat([ya, yb], 1), B, 32)
    scale = ref[..., 1].abs().max().item() + 1.0
    assert (got.double() - ref).abs().max().item() <= 2e-5 * scale


def test_colsums_refused_where_unsupported(device):
    """GEGLU and fp32 outputs cannot leave column sums: the wrapper reports None (the engine then runs groupnorm_stats), and
    a descriptor that forces them fails loudly."""
    import ctypes as C
    from seervideoldm_amd import _lib, ops
    a = _rand((1024, 320), device, 1).to(bf16)
    w = (_rand((2560, 320), device, 2) / 18).to(bf16)
    y = ops.gemm(a, w, geglu=True, colsum_batch=2)
    assert y.colsums is None
    # a split-K launch leaves them through its reduce pass; a reused `out=` tensor never keeps the sums of an earlier launch
    xs = _rand((4 * 8 * 8, 1280), device, 7).to(bf16)
    ws_ = (_rand((1280, 9 * 1280), device, 8) / 107).to(bf16)
    ys = ops.conv3x3(xs, ws_, 4, 8, 8, colsum_batch=2)
    assert ys.colsums is not None
    ys2 = ops.conv3x3(xs, ws_, 4, 8, 8, out=ys)
    assert ys2 is ys and ys.colsums is None
    y = ops.gemm(a, w[:320].contiguous(), out_f32=True, colsum_batch=2)
    assert y.colsums is None
    d = _lib.GemmDesc()
    out = torch.empty((1024, 1280), device=device, dtype=bf16)
    buf = torch.empty((1 << 20,), device=device, dtype=torch.float32)
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K, d.K1, d.lda, d.ldc, d.batch = 1024, 2560, 320, 320, 320, 1280, 1
    d.epilogue = _lib.SEER_EPI_GEGLU
    d.colsum = buf.data_ptr()
    lib = _lib.load()
    assert lib.seer_gemm_colsum_rows(C.byref(d)) == 0
    assert lib.seer_gemm_bf16(C.byref(d), torch.cuda.current_stream().cuda_stream) != 0


@pytest.mark.parametrize("rows,C", [(1000, 320), (513, 640), (77, 1280)])
def test_layernorm(device, rows, C):
    from seervideoldm_amd import ops
    x = (_rand((rows, C), device, 1) * 3 + 1).to(bf16)
    gamma = _rand((C,), device, 2) + 1.0
    beta = _rand((C,), device, 3)
    y = ops.layernorm(x, gamma, beta)
    _close(y, Fn.layer_norm(x.float(), (C,), gamma, beta), what="layernorm")


def test_softmax_rows(device):
    from seervideoldm_amd import ops
    x = (_rand((300, 1024), device, 1) * 4).to(bf16)
    y = ops.softmax_rows(x, 0.37)
    _close(ops.softmax_rows(x.float(), 0.37), (x.float() * 0.37).softmax(-1), rtol=2e-2, atol=1e-4, what="softmax f32 in")
    _close(y, (x.float() * 0.37).softmax(-1), rtol=2e-2, atol=1e-4, what="softmax")


# ---------------------------------------------------------------------------------------------------- small kernels
def test_timestep_embedding_and_small_linear(device):
    from seervideoldm_amd import ops
    t = torch.tensor([751, 1], device=device, dtype=torch.int64)
    emb = ops.timestep_embedding(t, 320, True, 0.0)
    half = 160
    expo = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=device) / half
    arg = t[:, None].float() * expo.exp()[None]
    ref = torch.cat([arg.cos(), arg.sin()], -1)
    _close(emb, ref, rtol=0, atol=2e-4, what="timestep embedding")
    w = _rand((1280, 320), device, 1, 0.05).to(bf16)
    b = _rand((1280,), device, 2)
    y = ops.linear_smallm(emb, w, b, silu_out=True)
    _close(y, Fn.silu(emb @ w.float().t() + b), rtol=1e-3, atol=1e-3, what="small linear")
    y2 = ops.linear_smallm(y, _rand((777, 1280), device, 3, 0.03).to(bf16), None, silu_in=True)
    _close(y2, Fn.silu(y) @ _rand((777, 1280), device, 3, 0.03).to(bf16).float().t(), rtol=1e-3, atol=1e-3,
           what="small linear silu_in")


def test_conv_in_out(device):
    from seervideoldm_amd import ops
    B, Fr, H, W = 2, 3, 16, 16
    x = _rand((B, 4, Fr, H, W), device, 1)
    w = _rand((320, 4, 3, 3), device, 2, 0.2)
    bias = _rand((320,), device, 3)
    y = ops.conv_in(x, w.permute(2, 3, 1, 0).contiguous(), bias)
    x2 = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, 4, H, W)
    ref = Fn.conv2d(x2, w, bias, padding=1).permute(0, 2, 3, 1).reshape(-1, 320)
    _close(y, ref, what="conv_in")
    wo = _rand((4, 320, 3, 3), device, 4, 0.02)
    bo = _rand((4,), device, 5)
    z = ops.conv_out(y, wo.permute(0, 2, 3, 1).contiguous(), bo, B, Fr, H, W)
    yin = y.float().reshape(B * Fr, H, W, 320).permute(0, 3, 1, 2)
    refo = Fn.conv2d(yin, wo, bo, padding=1).reshape(B, Fr, 4, H, W).permute(0, 2, 1, 3, 4)
    _close(z, refo, rtol=1e-3, atol=1e-3, what="conv_out")
    # MFMA route (bf16 weights in conv3x3 packing, batched implicit GEMM with a transposed fp32 store)
    from seervideoldm_amd.weights import pack_conv3x3
    wob = pack_conv3x3(wo).to(bf16).contiguous()
    zm = ops.conv_out(y, wob, bo, B, Fr, H, W)
    refm = Fn.conv2d(yin, wob.float().reshape(4, 3, 3, 320).permute(0, 3, 1, 2), bo, padding=1)
    refm = refm.reshape(B, Fr, 4, H, W).permute(0, 2, 1, 3, 4)
    _close(zm, refm, rtol=1e-3, atol=1e-3, what="conv_out (MFMA)")
    # odd pixel counts / a second shape for the LDS-staged conv_in
    x3 = _rand((1, 4, 2, 8, 24), device, 7)
    y3 = ops.conv_in(x3, w.permute(2, 3, 1, 0).contiguous(), None)
    ref3 = Fn.conv2d(x3.permute(0, 2, 1, 3, 4).reshape(2, 4, 8, 24), w, None, padding=1).permute(0, 2, 3, 1).reshape(-1, 320)
    _close(y3, ref3, what="conv_in (no bias, 8x24)")


def test_layout_and_cast(device):
    from seervideoldm_amd import ops
    x = _rand((3, 37, 5, 9), device, 1)
    y = ops.nchw_to_nhwc_bf16(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).reshape(-1, 37).to(bf16))
    z = ops.nhwc_to_nchw_f32(y, 3, 5, 9)
    assert torch.equal(z, x.to(bf16).float())
    c = _rand((1001,), device, 2)
    assert torch.equal(ops.cast_bf16(c), c.to(bf16))


def test_cfg_ddim_step(device):
    from seervideoldm_amd import ops
    b, C, Fp, cf, h, w = 2, 4, 5, 2, 8, 8
    eps = _rand((2 * b, C, Fp + cf, h, w), device, 1)
    x = _rand((b, C, Fp, h, w), device, 2)
    noise = _rand((b, C, Fp, h, w), device, 3)
    coef = torch.tensor([[0.5, 0.7, 0.1, math.sqrt(0.5)], [0.0376981, 0.329366, 0.0, math.sqrt(1 - 0.0376981)]],
                        device=device)
    for idx, nz in ((0, noise), (1, None)):
        xp, x0 = ops.cfg_ddim_step(eps, x, coef, idx, cfg=True, scale=7.5, cond_f=cf, noise=nz)
        eu, ec = eps.chunk(2)
        e = eu[:, :, cf:] + 7.5 * (ec[:, :, cf:] - eu[:, :, cf:])
        a_t, a_p, sg, s1 = coef[idx].tolist()
        rx0 = (x - s1 * e) / math.sqrt(a_t)
        rxp = math.sqrt(a_p) * rx0 + math.sqrt(1 - a_p - sg * sg) * e + (sg * noise if nz is not None else 0)
        _close(x0, rx0, rtol=1e-5, atol=1e-5, what="pred_x0")
        _close(xp, rxp, rtol=1e-5, atol=1e-5, what="x_prev")
    img = _rand((1000,), device, 4) * 2
    ref = ((img + 1) / 2).clamp(0, 1)
    assert torch.allclose(ops.clamp01_(img.clone()), ref)


@pytest.mark.parametrize("C1,C2,rows,tile", [(320, 0, 1536, 0), (640, 0, 768, 0), (1280, 0, 512, 0), (640, 320, 768, 0), (1280, 1280, 256, 0),
                                             (320, 320, 1536, 0), (1280, 640, 512, 0), (320, 0, 2048, 22), (1280, 0, 1024, 22),
                                             (640, 0, 3072, 0), (640, 640, 3072, 0),
                                             # slice widths 64 and 72 (16 and 14 tile lanes in the statistics pass: past the 12 the
                                             # UNet's own widths use), group widths 16 / 32 / 64 / 24
                                             (512, 0, 768, 0), (1024, 0, 512, 0), (2048, 0, 256, 0), (768, 0, 512, 0), (512, 512, 512, 0)])
def test_groupnorm_apply_from_colsums(device, C1, C2, rows, tile):
    """One launch = seer_groupnorm_stats_from_colsums + seer_groupnorm_apply: every block re-derives the statistics of its own
    groups from the producers' column sums (resnet.py:179,197 / attention.py:133 normalise the 5-D tensor per (batch, group)).
    Against F.group_norm in fp32 on the stored bf16 activations and against the two-launch form (a different fp32 order of the
    same additions: equal to rounding); channel widths incl. both skip-concat layouts (group width 30 and 60), column sums from the
    64-row partials of the 256 x 320 tile."""
    from seervideoldm_amd import ops
    B, G = 2, 32
    M = B * rows

    def produce(C, seed):
        a = _rand((M, 320), device, seed).to(bf16)
        w = _rand((C, 320), device, seed + 1, 320 ** -0.5).to(bf16)
        y = ops.gemm(a, w, bias=_rand((C,), device, seed + 2), tile=tile, splits=1 if tile else 0, colsum_batch=B)
        assert y.colsums is not None
        return y
    x1 = produce(C1, 1)
    x2 = produce(C2, 11) if C2 else None
    C = C1 + C2
    gamma, beta = _rand((C,), device, 21) + 1.0, _rand((C,), device, 22)
    count = rows * (C // G)
    got = ops.groupnorm_apply_from_colsums(x1, x2, x1.colsums, x2.colsums if C2 else None, B, G, count, 1e-5, gamma, beta, True)
    assert got is not None, "these layouts slice into whole groups of 64..128 channels and are small enough for the one-launch form"
    stats = torch.zeros((B, G, 2), device=device, dtype=torch.float32)
    ops.groupnorm_stats_from_colsums(x1.colsums, x2.colsums if C2 else None, B, G, stats)
    two = ops.groupnorm_apply(x1, x2, B, G, stats, count, 1e-5, gamma, beta, True)
    xc = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    ref = Fn.silu(Fn.group_norm(xc.reshape(B, rows, C).permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(M, C)
    _close(got, ref, rtol=1e-2, atol=1e-2, what=f"fused groupnorm C {C1}+{C2}")
    assert (got.float() - two.float()).abs().max().item() <= 2 ** -6 * ref.abs().max().item()      # at most a bf16 ulp apart
    again = ops.groupnorm_apply_from_colsums(x1, x2, x1.colsums, x2.colsums if C2 else None, B, G, count, 1e-5, gamma, beta, True)
    assert torch.equal(got, again), "fixed order of additions: bit-identical from launch to launch"


@pytest.mark.parametrize("kind,shape,tile,splits", [
    ("gemm", (1536, 640, 640), 0, 0), ("gemm", (3072, 320, 320), 16, 1), ("gemm", (768, 1280, 2560), 0, 0),
    ("gemm", (2048, 320, 2560), 22, 1), ("gemm", (1024, 1280, 5120), 22, 4),
    ("conv", (8, 16, 16, 640, 640), 0, 0), ("conv", (24, 4, 4, 1280, 1280), 0, 0), ("conv", (8, 16, 16, 640, 640), 8, 1),
    ("conv", (8, 8, 8, 1280, 1280), 5, 4), ("conv", (8, 16, 16, 640, 640), 22, 1), ("up", (4, 8, 8, 640, 640), 0, 0),
    ("up", (4, 8, 8, 640, 640), 22, 1), ("down", (8, 16, 16, 320, 320), 0, 0),
])
def test_column_sums_accumulated_in_fixed_point(device, kind, shape, tile, splits):
    """seer_gemm_desc::colsum_fx: the launch ADDS (sum, sum of squares) of its stored output per (batch element, column) to an
    int64 buffer at scale 2^20 -- tile epilogue, split-K reduce pass, the four phases of an upsampling conv, the 256 x 320 tile.
    Against torch sums in fp64 of the stored bf16 output (resnet.py:179,197 / attention.py:133 take the statistics of exactly
    these tensors); integer adds commute: a second launch into a fresh buffer gives the same bits."""
    from seervideoldm_amd import ops
    B = 2

    def run(arena):
        cb = (B, arena)
        if kind == "gemm":
            M, N, K = shape
            a = _rand((M, K), device, 1).to(bf16)
            w = (_rand((N, K), device, 2) / math.sqrt(K)).to(bf16)
            return ops.gemm(a, w, bias=_rand((N,), device, 5), residual=_rand((M, N), device, 3).to(bf16), tile=tile, splits=splits,
                            colsum_batch=cb)
        n_img, H, W, Ci, Co = shape
        x = _rand((n_img * H * W, Ci), device, 1).to(bf16)
        if kind == "up":
            from seervideoldm_amd.weights import pack_conv3x3_up_phases
            w = _rand((Co, Ci, 3, 3), device, 2) / math.sqrt(9 * Ci)
            return ops.conv_up2x(x, pack_conv3x3_up_phases(w).to(bf16), n_img, H, W, bias=_rand((Co,), device, 5), tile=tile, colsum_batch=cb)
        w = (_rand((Co, 9 * Ci), device, 2) / math.sqrt(9 * Ci)).to(bf16)
        if kind == "down":
            return ops.conv3x3(x, w, n_img, H, W, stride=2, bias=_rand((Co,), device, 5), tile=tile, splits=splits, colsum_batch=cb)
        return ops.conv3x3(x, w, n_img, H, W, bias=_rand((Co,), device, 5), rowvec=_rand((B, Co), device, 4),
                           rows_per_batch=n_img // B * H * W, tile=tile, splits=splits, colsum_batch=cb)
    arena = ops.FxArena(device, 1 << 18)
    arena.reset()
    y = run(arena)
    fx = y.colsums
    assert isinstance(fx, ops.ColSumsFx), "this launch was expected to accumulate its column sums"
    v = y.double().reshape(B, -1, y.shape[1])
    ref = torch.stack([v.sum(dim=1), (v * v).sum(dim=1)], -1)
    got = fx.totals()
    # one rounding to 2^-21 per partial, at most rows / 4 partials per column; the partial itself is an fp32 sum of <= 256 terms
    tol = 2.0 ** -21 * v.shape[1] / 4 + 1e-6 * (v.abs().sum(dim=1).max().item() + (v * v).sum(dim=1).max().item())
    assert (got - ref).abs().max().item() <= tol, ((got - ref).abs().max().item(), tol)
    arena2 = ops.FxArena(device, 1 << 18)
    arena2.reset()
    y2 = run(arena2)
    assert torch.equal(y2, y) and torch.equal(y2.colsums.buf, fx.buf), "integer accumulation: the same bits from launch to launch"
    arena.reset()
    assert int(arena.buf.abs().max().item()) == 0 and arena.used == 0


@pytest.mark.parametrize("C1,C2,rows", [(320, 0, 1536), (640, 0, 768), (1280, 0, 512), (640, 320, 768), (1280, 1280, 256),
                                        (320, 320, 12288), (1280, 640, 512), (640, 640, 3072), (2560, 0, 192)])
def test_groupnorm_apply_fx(device, C1, C2, rows):
    """GroupNorm (+ SiLU) whose statistics are the producers' accumulated fixed-point column sums: ONE launch per norm, against
    F.group_norm in fp32 on the stored bf16 activations (resnet.py:179,197, attention.py:133) and against the two-launch form;
    both skip-concat layouts (group width 30 and 60)."""
    from seervideoldm_amd import ops
    B, G = 2, 32
    M = B * rows
    arena = ops.FxArena(device, 1 << 18)
    arena.reset()

    def produce(C, seed):
        a = _rand((M, 320), device, seed).to(bf16)
        w = _rand((C, 320), device, seed + 1, 320 ** -0.5).to(bf16)
        y = ops.gemm(a, w, bias=_rand((C,), device, seed + 2) * 3, colsum_batch=(B, arena))
        assert isinstance(y.colsums, ops.ColSumsFx)
        return y
    x1 = produce(C1, 1)
    x2 = produce(C2, 11) if C2 else None
    C = C1 + C2
    gamma, beta = _rand((C,), device, 21) + 1.0, _rand((C,), device, 22)
    count = rows * (C // G)
    st_out = torch.zeros((B, G, 2), device=device, dtype=torch.float32)
    got = ops.groupnorm_apply_fx(x1, x2, x1.colsums, x2.colsums if C2 else None, B, G, count, 1e-5, gamma, beta, True, stats_out=st_out)
    assert got is not None
    xc = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    ref = Fn.silu(Fn.group_norm(xc.reshape(B, rows, C).permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(M, C)
    _close(got, ref, rtol=1e-2, atol=1e-2, what=f"groupnorm_apply_fx C {C1}+{C2}")
    stats = torch.zeros((B, G, 2), device=device, dtype=torch.float32)
    ops.groupnorm_stats(x1, x2, B, G, stats)
    two = ops.groupnorm_apply(x1, x2, B, G, stats, count, 1e-5, gamma, beta, True)
    assert (got.float() - two.float()).abs().max().item() <= 2 ** -6 * ref.abs().max().item()      # at most a bf16 ulp apart
    s2 = torch.zeros_like(stats)
    ops.groupnorm_stats_from_fx(x1.colsums, x2.colsums if C2 else None, B, G, s2)
    assert (s2 - stats).abs().max().item() <= 1e-4 * stats.abs().max().item()
    assert (st_out - stats).abs().max().item() <= 1e-4 * stats.abs().max().item(), "the statistics the backward pass takes"


@pytest.mark.parametrize("M,N,K,tile,res", [(1536, 320, 320, 0, True), (3072, 640, 640, 16, True), (768, 1280, 1280, 7, False),
                                            (384, 1280, 1280, 0, True), (1536, 320, 1280, 8, True), (2048, 640, 2560, 5, True),
                                            (1000, 320, 320, 0, True), (1536, 328, 320, 8, False)])
def test_gemm_row_statistics(device, M, N, K, tile, res):
    """seer_gemm_desc::rowstat: (sum, sum of squares) of every stored output row per tile column -- what the LayerNorm folded into
    the consuming GEMM normalises with (attention.py:198-200, 275-277 run nn.LayerNorm on exactly these rows); ragged M and N."""
    from seervideoldm_amd import ops
    a = _rand((M, K), device, 1).to(bf16)
    w = _rand((N, K), device, 2, K ** -0.5).to(bf16)
    y = ops.gemm(a, w, bias=_rand((N,), device, 3), residual=_rand((M, N), device, 4).to(bf16) if res else None, tile=tile, rowstat=True)
    rs = y.rowstats
    assert rs is not None and rs.buf.shape == (M, 2)
    got = rs.totals()
    v = y.double()
    ref = torch.stack([v.sum(dim=1), (v * v).sum(dim=1)], -1)
    # the sums are taken from the fp32 values the epilogue rounds to bf16: 2^-9 relative per element, adding up like a random walk
    tol = 2.0 ** -9 * (v * v).sum(dim=1).sqrt().max().item() * 4 + 2.0 ** -8 * ref[:, 1].max().item() / math.sqrt(N) * 4
    assert (got - ref).abs().max().item() <= tol, ((got - ref).abs().max().item(), tol)
    plain = ops.gemm(a, w, bias=_rand((N,), device, 3), residual=_rand((M, N), device, 4).to(bf16) if res else None, tile=tile)
    assert torch.equal(plain, y), "the statistics are a side output: same bits in C"


@pytest.mark.parametrize("C,M,kind,ptile,ctile", [
    (320, 1536, "qkv", 0, 0), (320, 3072, "geglu", 16, 5), (640, 1536, "qkv", 8, 16), (640, 768, "geglu", 0, 0),
    (1280, 768, "q", 7, 7), (1280, 384, "geglu", 0, 0), (320, 2048, "rotary", 0, 0), (1280, 384, "rotary", 0, 8),
    (640, 1000, "qkv", 0, 0), (320, 24576, "qkv", 0, 0),
])
def test_layernorm_folded_into_gemm(device, C, M, kind, ptile, ctile):
    """LayerNorm + Linear as ONE GEMM over the un-normalised rows (seer_gemm_desc::ln_rowstat; norm1 -> to_q|k|v, norm2 ->
    attn2.to_q, norm3 -> ff.net.0 of attention.py:231-246, 308-327): against nn.LayerNorm + nn.Linear in fp32 on the stored bf16
    rows, and against the two-launch form (layernorm kernel + plain weights); rows with a large common offset (mean >> std)."""
    from seervideoldm_amd import ops
    Hh = 8
    d = C // Hh
    a = _rand((M, C), device, 1).to(bf16)
    wp = _rand((C, C), device, 2, C ** -0.5).to(bf16)
    x = ops.gemm(a, wp, bias=_rand((C,), device, 3) * 2 + 1.5, residual=(_rand((M, C), device, 4) * 2).to(bf16), tile=ptile, rowstat=True)
    assert x.rowstats is not None
    gamma, beta = _rand((C,), device, 5) * 0.3 + 1.0, _rand((C,), device, 6) * 0.2
    N = {"qkv": 3 * C, "q": C, "geglu": 8 * C, "rotary": 3 * C}[kind]
    w = _rand((N, C), device, 7, C ** -0.5).to(bf16)
    bias = _rand((N,), device, 8) if kind == "geglu" else None
    kw = {}
    if kind in ("qkv", "q"):
        kw["col_scale"] = (0.7, C)
    if kind == "geglu":
        kw["geglu"] = True
    if kind == "rotary":
        T = M // 2
        freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(device)
        cs = ops.rotary_table(freqs, T)
        kw["rotary"] = (cs, T, 0, d, min(32, d), 2 * C)
        kw["col_scale"] = (0.7, C)
    wf, wsum, bf = ops.fold_layernorm(w, gamma, beta, bias)
    got = ops.gemm(x, wf, bias=bf, ln=(x.rowstats, wsum, 1e-5), tile=ctile, **kw)
    assert got is not None, "this launch was expected to fold the LayerNorm"
    n = ops.layernorm(x, gamma, beta)
    two = ops.gemm(n, w, bias=bias, tile=ctile, **kw)
    # fp32 reference of the same stages
    ln = Fn.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    acc = ln @ w.float().t()
    if bias is not None:
        acc = acc + bias
    if kind == "geglu":
        # value / gate column pairs interleave in blocks of 16 inside the packed weight (weights.pack_geglu): compare against the
        # two-launch form only (it reads the same packing)
        ref = two.float()
    else:
        if kind == "rotary":
            pos = torch.arange(M, device=device) % (M // 2)
            c, s = cs[pos, :, 0], cs[pos, :, 1]
            rd = min(32, d)
            t = acc[:, :2 * C].reshape(M, 2 * Hh, d)
            x0, x1 = t[..., :rd:2], t[..., 1:rd:2]
            rot = torch.stack([x0 * c[:, None] - x1 * s[:, None], x1 * c[:, None] + x0 * s[:, None]], -1).flatten(-2)
            acc = torch.cat([torch.cat([rot, t[..., rd:]], -1).reshape(M, 2 * C), acc[:, 2 * C:]], 1)
        acc = torch.cat([acc[:, :C] * 0.7, acc[:, C:]], 1)
        ref = acc
    if kind == "geglu":
        # value * gelu(gate) against ANOTHER bf16 result (not fp32): a gate off by one bf16 step moves a product with a large
        # value by more than an element-wise tolerance allows; bound the error against the tensor's scale instead
        e = (got.float() - ref)
        assert torch.isfinite(got.float()).all()
        assert e.pow(2).mean().sqrt().item() <= 1e-2 * ref.pow(2).mean().sqrt().item()
        assert e.abs().max().item() <= 1e-2 * ref.abs().max().item()
    else:
        _close(got, ref, rtol=2e-2, atol=2e-2, what=f"folded LayerNorm {kind} C{C}")
    # ... and no further from the fp32 reference than the two-launch form is (it rounds LN(x) to bf16; this one rounds gamma (.) W)
    if kind != "geglu":
        e_fold = (got.float() - ref).pow(2).mean().sqrt().item()
        e_two = (two.float() - ref).pow(2).mean().sqrt().item()
        assert e_fold <= 1.5 * e_two + 1e-4, (e_fold, e_two)
    again = ops.gemm(x, wf, bias=bf, ln=(x.rowstats, wsum, 1e-5), tile=ctile, **kw)
    assert torch.equal(got, again)


@pytest.mark.parametrize("C,rows,B", [(320, 3072, 2), (640, 768, 1), (1280, 192, 2), (2560, 48, 1), (320, 12288, 1), (72, 100, 3)])
def test_groupnorm_stats_fx_is_exact_and_shard_invariant(device, C, rows, B):
    """seer_groupnorm_stats_fx: the accumulated fixed-point (sum, sum of squares) per (batch element, column) from the activations.
    Every element is rounded on its own, so the totals are integer sums: EQUAL to the torch int64 reference, independent of the block
    decomposition, and the sums of two row shards (what frame-sharded ranks all-reduce) ARE the unsharded sums -- and
    seer_groupnorm_apply_fx normalises identically from either (resnet.py:179,197: GroupNorm statistics span all frames)."""
    from seervideoldm_amd import ops
    x = (_rand((B * rows, C), device, 5) * 3 + 0.7).to(bf16)
    fx = ops.groupnorm_stats_fx(x, B)
    v = x.float().reshape(B, rows, C)
    ref = torch.stack([torch.round(v * 2.0 ** 20).to(torch.int64).sum(1), torch.round(v * v * 2.0 ** 20).to(torch.int64).sum(1)], 1)
    assert torch.equal(fx.buf[0], ref)
    assert torch.equal(ops.groupnorm_stats_fx(x, B).buf, fx.buf)
    # two "frame shards" of every batch element, uneven
    cut = (rows * 2 // 3) // 4 * 4 or rows // 2
    xs = x.reshape(B, rows, C)
    a, b = xs[:, :cut].reshape(-1, C).contiguous(), xs[:, cut:].reshape(-1, C).contiguous()
    fa, fb = ops.groupnorm_stats_fx(a, B), ops.groupnorm_stats_fx(b, B)
    assert torch.equal(fa.buf + fb.buf, fx.buf)
    if C % 32 == 0 and C >= 320:
        G = 32
        gamma, beta = _rand((C,), device, 21) + 1.0, _rand((C,), device, 22)
        count = rows * (C // G)
        y_full = ops.groupnorm_apply_fx(x, None, fx, None, B, G, count, 1e-5, gamma, beta, True)
        assert y_full is not None
        tot = ops.ColSumsFx(fa.buf + fb.buf, C)
        y_a = ops.groupnorm_apply_fx(a, None, tot, None, B, G, count, 1e-5, gamma, beta, True)
        y_b = ops.groupnorm_apply_fx(b, None, tot, None, B, G, count, 1e-5, gamma, beta, True)
        got = torch.cat([y_a.reshape(B, cut, C), y_b.reshape(B, rows - cut, C)], 1).reshape(-1, C)
        assert torch.equal(got, y_full), "a shard normalises its rows exactly as the unsharded launch does"
        refy = Fn.silu(Fn.group_norm(v.permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(-1, C)
        _close(y_full, refy, rtol=1e-2, atol=1e-2, what=f"groupnorm from exact sums C {C}")


# ------------------------------------------------------------------------------ fused feed-forward, 320 channels
@pytest.mark.parametrize("M,B,strided", [(96, 1, False), (960, 2, False), (24576, 2, False), (12288, 1, True),
                                         (1000, 1, False),      # a ragged last tile (40 rows)
                                         (2048, 2, False),      # 1024 rows per batch element: tile 10 straddles the two
                                         (28672, 2, True)])     # config 2 at 14 frames: 298 tiles + 64 rows, the boundary inside tile 149
def test_ff_fused_c320(device, M, B, strided):
    """seer_ff_fused_c320: y = x + [Wp | Wp W2][h | GEGLU(LN(h) W1^T + b1)] + bcat in one launch, against (a) the fp32 formula on the
    bf16-rounded operands with the intermediate roundings of the unfused path (LN(h) and g stored as bf16) and (b) the three
    launches it replaces; the fixed-point column sums against the sums of the stored output."""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import geglu_row_order
    C, inner = 320, 1280
    ld = C + 64 if strided else C
    hbuf = _rand((M, ld), device, 1).to(bf16)
    xbuf = _rand((M, ld), device, 2).to(bf16)
    h, x = hbuf[:, :C], xbuf[:, :C]
    gamma, beta = 1.0 + 0.2 * _rand((C,), device, 3), 0.1 * _rand((C,), device, 4)
    w1 = _rand((2 * inner, C), device, 5, C ** -0.5).to(bf16)
    b1 = 0.2 * _rand((2 * inner,), device, 6)
    wcat = _rand((C, C + inner), device, 7, (C + inner) ** -0.5).to(bf16)
    bcat = 0.2 * _rand((C,), device, 8)
    order = geglu_row_order(inner, device)
    w1p, b1p = w1[order].contiguous(), b1[order].contiguous()
    w1f, wcf = ops.ff_fused_pack(w1p, wcat)
    arena = ops.FxArena(device, 1 << 16)
    arena.reset()
    y = ops.ff_fused(h, x, gamma, beta, w1f, b1p, wcf, bcat, colsum_batch=(B, arena))
    assert y is not None and y.shape == (M, C)
    torch.cuda.synchronize()
    # (a) the formula
    hn = Fn.layer_norm(h.float(), (C,), gamma, beta, 1e-5).to(bf16).float()
    pre = hn @ w1.float().t() + b1
    g = (pre[:, :inner] * Fn.gelu(pre[:, inner:])).to(bf16).float()
    ref = x.float() + torch.cat([h.float(), g], 1) @ wcat.float().t() + bcat
    _close(y, ref, what="ff_fused vs formula")
    rel = ((y.float() - ref).norm() / ref.norm()).item()
    assert rel < 4e-3, rel
    # (b) the launches it replaces
    n = ops.layernorm(h.contiguous(), gamma, beta)
    gg = ops.gemm(n, w1p, bias=b1p, geglu=True)
    y3 = ops.gemm(h.contiguous(), wcat, a2=gg, bias=bcat, residual=x.contiguous())
    rel3 = ((y.float() - y3.float()).norm() / y3.float().norm()).item()
    assert rel3 < 4e-3, rel3
    # column sums of the stored values (accumulated per batch element: its rows a multiple of 16)
    cs = y.colsums
    if (M // B) % 16:
        assert cs is None
        return
    assert cs is not None
    tot = cs.totals()                                                      # [B, C, 2] fp64
    yb = y.double().reshape(B, M // B, C)
    assert torch.allclose(tot[:, :, 0], yb.sum(1), rtol=0, atol=2e-2)
    assert torch.allclose(tot[:, :, 1], (yb * yb).sum(1), rtol=1e-5, atol=2e-2)
    # the per-tile form of the same sums (where no tile straddles two batch elements; else the launch leaves none)
    yt = ops.ff_fused(h, x, gamma, beta, w1f, b1p, wcf, bcat, colsum_batch=B)
    assert torch.equal(yt, y)
    if (M // B) % 96 == 0:
        assert isinstance(yt.colsums, ops.ColSums) and yt.colsums.tiles == M // 96
        tt = yt.colsums.buf.double().reshape(B, M // B // 96, C, 2).sum(1)
        assert torch.allclose(tt[:, :, 0], yb.sum(1), rtol=0, atol=2e-2) and torch.allclose(tt[:, :, 1], (yb * yb).sum(1), rtol=1e-5, atol=2e-2)
        stats = torch.empty((B, 32, 2), device=device)
        ops.groupnorm_stats_from_colsums(yt.colsums, None, B, 32, stats)
        want = torch.stack([yb.reshape(B, M // B, 32, 10).sum((1, 3)), (yb * yb).reshape(B, M // B, 32, 10).sum((1, 3))], -1)
        assert torch.allclose(stats.double(), want, rtol=1e-4, atol=1e-1)
    else:
        assert yt.colsums is None
    # in place on the residual stream
    x2 = x.contiguous().clone()
    y2 = ops.ff_fused(h, x2, gamma, beta, w1f, b1p, wcf, bcat, out=x2)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("dt,M,strided", [(bf16, 24576, False), (bf16, 1000, True), (torch.float16, 6144, False)])
def test_ff_fused_c320_with_the_to_out_prologue(device, dt, M, strided):
    """seer_ff_fused_c320_pre: the rows the fused feed-forward reads as h are h + a Wo^T + bo -- the attention's to_out projection and
    its residual (attention.py:237-240, 316-322) -- computed in the launch's tile and stored nowhere; against the two launches
    (to_out + residual, then the fused feed-forward) and the fp32 formula"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import geglu_row_order
    C, inner = 320, 1280
    ld = C + 64 if strided else C
    a = _rand((M, ld), device, 11).to(dt)[:, :C]
    h0 = _rand((M, ld), device, 1).to(dt)[:, :C]
    x = _rand((M, C), device, 2).to(dt)
    wo, bo = _rand((C, C), device, 12, C ** -0.5).to(dt), 0.1 * _rand((C,), device, 13)
    gamma, beta = 1.0 + 0.2 * _rand((C,), device, 3), 0.1 * _rand((C,), device, 4)
    w1 = _rand((2 * inner, C), device, 5, C ** -0.5).to(dt)
    b1 = 0.2 * _rand((2 * inner,), device, 6)
    wcat = _rand((C, C + inner), device, 7, (C + inner) ** -0.5).to(dt)
    bcat = 0.2 * _rand((C,), device, 8)
    order = geglu_row_order(inner, device)
    w1p, b1p = w1[order].contiguous(), b1[order].contiguous()
    w1f, wcf = ops.ff_fused_pack(w1p, wcat)
    h_keep = h0.clone()
    y = ops.ff_fused(h0, x, gamma, beta, w1f, b1p, wcf, bcat, pre=(a, ops.rowchain_pack(wo), bo))
    assert y is not None and y.dtype == dt and torch.equal(h0, h_keep), "the residual stream is read, not written"
    # the two launches it replaces: same roundings at the same places (h rounded to 16 bits once)
    h2 = ops.gemm(a.contiguous(), wo, bias=bo, residual=h0.contiguous())
    y2 = ops.ff_fused(h2, x, gamma, beta, w1f, b1p, wcf, bcat)
    tol = 4e-3 if dt == bf16 else 6e-4
    rel2 = ((y.float() - y2.float()).norm() / y2.float().norm()).item()
    assert rel2 < tol, rel2
    # the formula
    h = (h0.float() + a.float() @ wo.float().t() + bo).to(dt).float()
    hn = Fn.layer_norm(h, (C,), gamma, beta, 1e-5).to(dt).float()
    pre = hn @ w1.float().t() + b1
    g = (pre[:, :inner] * Fn.gelu(pre[:, inner:])).to(dt).float()
    ref = x.float() + torch.cat([h, g], 1) @ wcat.float().t() + bcat
    rel = ((y.float() - ref).norm() / ref.norm()).item()
    assert rel < tol, rel
    assert torch.equal(ops.ff_fused(h0, x, gamma, beta, w1f, b1p, wcf, bcat, pre=(a, ops.rowchain_pack(wo), bo)), y)


def test_ff_fused_c320_refuses_other_widths_and_the_engine_takes_it_where_it_pays(device):
    from seervideoldm_amd import ops
    z = torch.zeros((96, 640), device=device, dtype=bf16)
    f = torch.zeros((2560,), device=device)
    assert ops.ff_fused(z, z, f[:640], f[:640], torch.zeros((2560, 320), device=device, dtype=bf16), f,
                        torch.zeros((320, 1600), device=device, dtype=bf16), f[:320]) is None
    # rounds x 74 us: config 2 (one round), config 4 and the bridge configs (well-filled rounds) yes; a CFG half per rank, config 2 at
    # 14 frames (299 workgroups: a second round for 43 of them) no
    assert ops.ff_fused_pays(24576) and ops.ff_fused_pays(98304) and ops.ff_fused_pays(131072) and ops.ff_fused_pays(139264)
    assert not ops.ff_fused_pays(12288) and not ops.ff_fused_pays(28672) and not ops.ff_fused_pays(6144)
