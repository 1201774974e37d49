import sys, torch
sys.path.insert(0, '/root/repo')
from seervideoldm_amd import ops
from seervideoldm_amd.weights import pack_conv3x3
dev = torch.device('cuda:0'); bf16 = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
torch.manual_seed(0)
# correctness
for tile in (16, 17, 18):
    for (M, N, K) in [(24576, 320, 1280), (1000, 320, 320), (130, 68, 192), (12288, 640, 640)]:
        a = torch.randn((M, K), device=dev).to(bf16); w = (torch.randn((N, K), device=dev) * K ** -0.5).to(bf16)
        bias = torch.randn((N,), device=dev); res = torch.randn((M, N), device=dev).to(bf16)
        out = ops.gemm(a, w, bias=bias, residual=res, tile=tile)
        ref = a.float() @ w.float().t() + bias + res.float()
        err = ((out.float() - ref).norm() / ref.norm()).item()
        assert err < 5e-3, (tile, M, N, K, err)
    x = torch.randn((24 * 32 * 32, 320), device=dev).to(bf16)
    wc = (torch.randn((320, 320, 3, 3), device=dev) * (9 * 320) ** -0.5)
    out = ops.conv3x3(x, pack_conv3x3(wc).to(bf16), 24, 32, 32, tile=tile)
    ref = torch.nn.functional.conv2d(x.float().reshape(24, 32, 32, 320).permute(0, 3, 1, 2), wc.to(bf16).float(), padding=1).permute(0, 2, 3, 1).reshape(-1, 320)
    err = ((out.float() - ref).norm() / ref.norm()).item()
    assert err < 5e-3, (tile, "conv", err)
print("correct")
for nimg in (24, 12):
    x = torch.randn((nimg * 32 * 32, 320), device=dev).to(bf16)
    w = pack_conv3x3(torch.randn((320, 320, 3, 3), device=dev) * 0.02).to(bf16)
    b = torch.randn((320,), device=dev); r = torch.randn((nimg * 1024, 320), device=dev).to(bf16)
    for tile in (0, 12, 16, 17, 18, 5):
        t = timeit(lambda: ops.conv3x3(x, w, nimg, 32, 32, bias=b, residual=r, tile=tile))
        print(f"conv n{nimg} 32x32 320->320 tile {tile:2d}: {t:7.1f} us  {2*nimg*1024*320*2880/t/1e6:6.1f} TF/s")
    a = torch.randn((nimg * 1024, 1280), device=dev).to(bf16); w2 = (torch.randn((320, 1280), device=dev) * 0.03).to(bf16)
    for tile in (0, 12, 16, 17, 18, 5):
        t = timeit(lambda: ops.gemm(a, w2, bias=b, residual=r, tile=tile))
        print(f"gemm M{nimg*1024} N320 K1280 tile {tile:2d}: {t:7.1f} us  {2*nimg*1024*320*1280/t/1e6:6.1f} TF/s")
    a3 = torch.randn((nimg * 1024, 320), device=dev).to(bf16); w3 = (torch.randn((320, 320), device=dev) * 0.05).to(bf16)
    for tile in (0, 12, 16, 17, 18, 2):
        t = timeit(lambda: ops.gemm(a3, w3, bias=b, residual=r, tile=tile))
        print(f"gemm M{nimg*1024} N320 K320  tile {tile:2d}: {t:7.1f} us  {2*nimg*1024*320*320/t/1e6:6.1f} TF/s")
