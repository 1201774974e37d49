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
