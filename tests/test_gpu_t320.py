"""The 256 x 320 tile GEMM / conv kernel (csrc/gemm_t320.hip, SEER_TILE_T256x320 = 22) against fp32 formulas of the operators
it replaces (nn.Linear: attention.py:484-489,742,783; InflatedConv3d: resnet.py:8-16,39,82,144,153), through the same
ops.gemm / ops.conv3x3 / ops.conv_up2x entry points as every other tile: unsplit and with K slices reduced inside the launch,
every epilogue term, ragged M, column sums, determinism, and the state of the counter buffer the split launches share."""
import math

import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16
T320 = 22


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _close(got, ref, rtol=2e-2, atol=2e-2, what=""):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside tolerance, max err {err.max().item():.4g}"


def _sync_is_zero(device):
    from seervideoldm_amd import ops
    buf = ops._sync_buffers.get((device.type, device.index if device.index is not None else torch.cuda.current_device()))
    return buf is None or int(buf.to(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize("M,N,K,splits", [
    (256, 320, 64, 1), (512, 640, 320, 1), (1536, 1280, 1280, 1), (6144, 640, 640, 1), (384, 320, 320, 1), (300, 960, 192, 1),
    (1536, 1280, 5120, 0), (1536, 1280, 5120, 2), (1536, 1280, 5120, 5), (1536, 1280, 5120, 16), (6144, 640, 2560, 3),
    (384, 1280, 5120, 7), (256, 320, 1024, 4),
])
def test_gemm_bias_residual(device, M, N, K, splits):
    """bias + residual (the to_out / ff.net.2 / proj_out launches), unsplit and with 2..16 K slices reduced inside the launch;
    M = 384 and 300 leave a ragged last row tile"""
    from seervideoldm_amd import ops
    a = _rand((M, K), device, 1).to(bf16)
    w = _rand((N, K), device, 2, K ** -0.5).to(bf16)
    bias = _rand((N,), device, 3)
    res = _rand((M, N), device, 4).to(bf16)
    out = ops.gemm(a, w, bias=bias, residual=res, tile=T320, splits=splits)
    _close(out, a.float() @ w.float().t() + bias + res.float(), what=f"t320 {M}x{N}x{K} splits {splits}")
    again = ops.gemm(a, w, bias=bias, residual=res, tile=T320, splits=splits)
    assert torch.equal(out, again), "slices are added in slice order: two launches agree bit for bit"
    assert _sync_is_zero(device), "a split launch leaves its counters zero"


def test_unsplit_matches_the_smaller_tiles_bit_for_bit(device):
    """same K order, same accumulation -> identical bits to the 128 x 128 tile on a full-tile shape"""
    from seervideoldm_amd import ops
    a = _rand((1536, 640), device, 1).to(bf16)
    w = _rand((1280, 640), device, 2, 640 ** -0.5).to(bf16)
    bias = _rand((1280,), device, 3)
    assert torch.equal(ops.gemm(a, w, bias=bias, tile=T320, splits=1), ops.gemm(a, w, bias=bias, tile=5, splits=1))


@pytest.mark.parametrize("M,C", [(512, 320), (1536, 640), (384, 1280)])
def test_geglu(device, M, C):
    """ff.net.0: GEGLU epilogue (attention.py:783-793), interleaved value / gate rows"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import interleave_geglu
    a = _rand((M, C), device, 1).to(bf16)
    w = _rand((8 * C, C), device, 2, C ** -0.5).to(bf16)
    b = _rand((8 * C,), device, 3, 0.5)
    wp, bp = interleave_geglu(w, b)
    out = ops.gemm(a, wp, bias=bp, geglu=True, tile=T320)
    h = a.float() @ w.float().t() + b
    val, gate = h.chunk(2, dim=-1)
    _close(out, val * Fn.gelu(gate), what=f"t320 geglu {M}x{C}")


@pytest.mark.parametrize("splits", [1, 4])
def test_dual_source_rowvec_colscale(device, splits):
    """the 1x1 shortcut over a skip concat (two K sources), the per-batch time-embedding row, and a scaled column range"""
    from seervideoldm_amd import ops
    M, K1, K2, N, B = 1536, 640, 1280, 640, 2
    a1 = _rand((M, K1), device, 1).to(bf16)
    a2 = _rand((M, K2), device, 2).to(bf16)
    w = _rand((N, K1 + K2), device, 3, (K1 + K2) ** -0.5).to(bf16)
    rv = _rand((B, N), device, 4)
    out = ops.gemm(a1, w, a2=a2, rowvec=rv, rows_per_batch=M // B, col_scale=(0.125, 320), tile=T320, splits=splits)
    ref = torch.cat([a1, a2], 1).float() @ w.float().t() + rv.repeat_interleave(M // B, 0)
    ref[:, :320] *= 0.125
    _close(out, ref, what=f"t320 dual source splits {splits}")


@pytest.mark.parametrize("d,T,splits", [(40, 768, 1), (80, 768, 1), (160, 256, 2)])
def test_rotary_epilogue(device, d, T, splits):
    """the temporal q|k|v projection: rotary on the q|k columns (attention.py:649-651) + the scaled q columns"""
    from seervideoldm_amd import ops
    heads, B = 8, 2
    Cq = heads * d
    M, K = B * T, Cq
    rd = 32
    a = _rand((M, K), device, 1).to(bf16)
    w = _rand((3 * Cq, K), device, 2, K ** -0.5).to(bf16)
    freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(device)
    table = ops.rotary_table(freqs, T)
    out = ops.gemm(a, w, rotary=(table, T, 0, d, rd, 2 * Cq), col_scale=(0.5, Cq), tile=T320, splits=splits)
    ref = ops.gemm(a, w, rotary=(table, T, 0, d, rd, 2 * Cq), col_scale=(0.5, Cq), tile=5, splits=1)
    _close(out, ref, rtol=1e-2, atol=1e-2, what=f"t320 rotary d{d}")


@pytest.mark.parametrize("n_img,H,W,Ci,Co,stride,splits", [
    (2, 16, 16, 64, 320, 1, 1), (4, 16, 16, 320, 320, 1, 1), (24, 8, 8, 640, 640, 1, 0), (24, 4, 4, 1280, 1280, 1, 0),
    (6, 16, 16, 640, 640, 1, 5), (2, 32, 32, 320, 320, 2, 1), (6, 16, 16, 640, 640, 2, 3), (2, 6, 10, 64, 320, 1, 1),
])
def test_conv3x3(device, n_img, H, W, Ci, Co, stride, splits):
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5).to(bf16)
    bias = _rand((Co,), device, 3)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = _rand((n_img * Ho * Wo, Co), device, 4).to(bf16)
    out = ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, stride=stride, bias=bias, residual=res, tile=T320, splits=splits)
    ref = Fn.conv2d(x.float(), w.float(), bias, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, Co) + res.float()
    _close(out, ref, what=f"t320 conv {Ci}->{Co} {H}x{W} s{stride} splits {splits}")
    assert _sync_is_zero(device)


def test_conv_pad_after_only(device):
    """the VAE encoder's Downsample: F.pad(x, (0, 1, 0, 1)) + conv(stride 2, padding 0) (ldm/modules/diffusionmodules/model.py:60-78)"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    n_img, H, W, Ci, Co = 2, 32, 32, 128, 320
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5).to(bf16)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    out = ops.conv3x3(x_cl, pack_conv3x3(w), n_img, H, W, stride=2, pad_after_only=True, tile=T320, splits=1)
    ref = Fn.conv2d(Fn.pad(x.float(), (0, 1, 0, 1)), w.float(), None, stride=2).permute(0, 2, 3, 1).reshape(-1, Co)
    _close(out, ref, what="t320 conv pad_after_only")


@pytest.mark.parametrize("n_img,H,W,Ci,Co", [(4, 8, 8, 640, 640), (2, 16, 16, 320, 320), (3, 6, 10, 64, 320)])
def test_conv_up2x_phases(device, n_img, H, W, Ci, Co):
    """Upsample3D (resnet.py:52-57) as four phase convs: the launch's grid.z"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3_up_phases
    x = _rand((n_img, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5)
    bias = _rand((Co,), device, 3)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    w4 = pack_conv3x3_up_phases(w).to(bf16)
    out = ops.conv_up2x(x_cl, w4, n_img, H, W, bias=bias, tile=T320)
    ref = ops.conv_up2x(x_cl, w4, n_img, H, W, bias=bias, tile=5)
    _close(out, ref, rtol=1e-2, atol=1e-2, what=f"t320 conv_up2x {Ci}->{Co}")
    full = Fn.conv2d(Fn.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.to(bf16).float(), bias, padding=1)
    _close(out, full.permute(0, 2, 3, 1).reshape(-1, Co), rtol=3e-2, atol=3e-2, what="t320 conv_up2x vs 9 taps")


@pytest.mark.parametrize("kind,shape,splits", [
    ("gemm", (1536, 640, 640), 1), ("gemm", (1536, 1280, 5120), 5), ("gemm", (1024, 320, 2560), 4),
    ("conv", (8, 16, 16, 640, 640), 1), ("conv", (8, 16, 16, 640, 640), 3), ("up", (4, 8, 8, 640, 640), 1),
])
def test_groupnorm_statistics_from_its_column_sums(device, kind, shape, splits):
    """the launch's 64-row column-sum partials give the GroupNorm statistics of its stored output (resnet.py:179,197)"""
    from seervideoldm_amd import ops
    B, G = 2, 32
    if kind == "gemm":
        M, N, K = shape
        a = _rand((M, K), device, 1).to(bf16)
        w = (_rand((N, K), device, 2) / math.sqrt(K)).to(bf16)
        y = ops.gemm(a, w, bias=_rand((N,), device, 5), residual=_rand((M, N), device, 3).to(bf16), tile=T320, splits=splits,
                     colsum_batch=B)
    elif kind == "conv":
        n_img, H, W, Ci, Co = shape
        x = _rand((n_img * H * W, Ci), device, 1).to(bf16)
        w = (_rand((Co, 9 * Ci), device, 2) / math.sqrt(9 * Ci)).to(bf16)
        y = ops.conv3x3(x, w, n_img, H, W, bias=_rand((Co,), device, 5), rowvec=_rand((B, Co), device, 4),
                        rows_per_batch=n_img // B * H * W, tile=T320, splits=splits, colsum_batch=B)
    else:
        from seervideoldm_amd.weights import pack_conv3x3_up_phases
        n_img, H, W, Ci, Co = shape
        x = _rand((n_img * H * W, Ci), device, 1).to(bf16)
        w = _rand((Co, Ci, 3, 3), device, 2) / math.sqrt(9 * Ci)
        y = ops.conv_up2x(x, pack_conv3x3_up_phases(w).to(bf16), n_img, H, W, bias=_rand((Co,), device, 5), tile=T320, colsum_batch=B)
    cs = y.colsums
    assert cs is not None, "this launch was expected to produce column sums"
    got = torch.zeros((B, G, 2), device=device, dtype=torch.float32)
    ops.groupnorm_stats_from_colsums(cs, None, B, G, got)
    v = y.double().reshape(B, -1, G, y.shape[1] // G)
    ref = torch.stack([v.sum(dim=(1, 3)), (v * v).sum(dim=(1, 3))], -1)
    scale = ref[..., 1].abs().max().item() + 1.0
    assert (got.double() - ref).abs().max().item() <= 2e-5 * scale, (got.double() - ref).abs().max().item()


def test_ragged_rows_refuse_column_sums(device):
    """M % 256 != 0: the 64-row partial layout would run past ceil(M / 64) -- the query says so and the launch is refused"""
    from seervideoldm_amd import ops
    a = _rand((384, 320), device, 1).to(bf16)
    w = _rand((320, 320), device, 2, 320 ** -0.5).to(bf16)
    y = ops.gemm(a, w, tile=T320, colsum_batch=2)
    assert y.colsums is None


def test_auto_routes_large_shapes_to_the_big_tile_and_agrees(device):
    """AUTO (tile 0) on a config-4-sized feed-forward projection and a long-K conv: whatever tile the cost model picks, the result
    is the operator's"""
    from seervideoldm_amd import ops
    from seervideoldm_amd.weights import pack_conv3x3
    a = _rand((24576, 640), device, 1).to(bf16)
    w = _rand((5120, 640), device, 2, 640 ** -0.5).to(bf16)
    out = ops.gemm(a, w)
    ref = ops.gemm(a, w, tile=5, splits=1)
    assert torch.equal(out, ref)            # unsplit big tile and 128 x 128 tile: the same K order
    n_img, H, Ci, Co = 24, 16, 1920, 640
    x = _rand((n_img, Ci, H, H), device, 3).to(bf16)
    wc = _rand((Co, Ci, 3, 3), device, 4, (9 * Ci) ** -0.5).to(bf16)
    x_cl = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    got = ops.conv3x3(x_cl, pack_conv3x3(wc), n_img, H, H)
    refc = Fn.conv2d(x.float(), wc.float(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    _close(got, refc, what="auto conv 16x16 1920->640")
    assert _sync_is_zero(device)
