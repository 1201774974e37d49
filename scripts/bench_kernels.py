"""Micro-benchmark of the hot kernels at the Sthv2 shapes: prints achieved TFLOP/s / GB/s.
Run on the GPU box:  python scripts/bench_kernels.py [gemm|conv|attn|norm|all]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402
from seervideoldm_amd.weights import interleave_geglu  # noqa: E402

bf16 = torch.bfloat16
dev = torch.device("cuda:0")


TILES = ((0, "auto"), (5, "g128x128/2"), (14, "g256x128/2"), (7, "g128x64/3"), (15, "g256x64/3"), (12, "g128x160/2"),
         (8, "g64x64/3"))


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def bench_gemm():
    shapes = [  # M, N, K, geglu
        (24576, 2560, 320, True), (6144, 5120, 640, True), (1536, 10240, 1280, True), (384, 10240, 1280, True),
        (24576, 320, 1280, False), (6144, 640, 2560, False), (1536, 1280, 5120, False), (384, 1280, 5120, False),
        (24576, 960, 320, False), (6144, 1920, 640, False), (1536, 3840, 1280, False),
        (24576, 320, 320, False), (6144, 640, 640, False), (1536, 1280, 1280, False), (384, 1280, 1280, False),
        (1848, 640, 768, False), (4096, 4096, 4096, False), (8192, 8192, 8192, False),
    ]
    for M, N, K, geglu in shapes:
        a = torch.randn(M, K, device=dev).to(bf16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf16)
        bias = torch.randn(N, device=dev)
        out = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=bf16)
        line = f"gemm M{M} N{N} K{K} geglu={int(geglu)}:"
        for tile, name in TILES:
            try:
                t = timeit(lambda: ops.gemm(a, w, bias=bias, geglu=geglu, out=out, tile=tile, splits=(0 if tile == 0 else 1)))
            except Exception:
                line += f"  {name}   n/a |"
                continue
            line += f"  {name} {2 * M * N * K / t / 1e12:5.0f}TF {t * 1e6:6.1f}us |"
        print(line, flush=True)


def bench_conv():
    shapes = [  # n_img, H, W, Ci, Co, stride, up
        (24, 32, 32, 320, 320, 1, False), (24, 16, 16, 640, 640, 1, False), (24, 8, 8, 1280, 1280, 1, False),
        (24, 4, 4, 1280, 1280, 1, False), (24, 8, 8, 2560, 1280, 1, False), (24, 16, 16, 1920, 640, 1, False),
        (24, 32, 32, 960, 320, 1, False), (24, 16, 16, 640, 640, 1, True), (24, 8, 8, 1280, 1280, 1, True),
        (24, 32, 32, 320, 320, 2, False),
    ]
    for n, H, W, Ci, Co, s, up in shapes:
        x = torch.randn(n * H * W, Ci, device=dev).to(bf16)
        w = (torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(bf16)
        bias = torch.randn(Co, device=dev)
        Ho = (2 * H if up else H) // s
        flops = 2 * n * Ho * Ho * Co * 9 * Ci
        line = f"conv n{n} {H}x{W} {Ci}->{Co} s{s} up{int(up)}:"
        for tile, name in TILES:
            t = timeit(lambda: ops.conv3x3(x, w, n, H, W, stride=s, upsample=up, bias=bias, tile=tile, splits=(0 if tile == 0 else 1)))
            line += f"  {name} {flops / t / 1e12:5.0f}TF {t * 1e6:6.1f}us |"
        print(line, flush=True)


def bench_split():
    """split-K sweep on the deep-level shapes (M = 384 / 1536)"""
    gemms = [(384, 1280, 1280), (384, 1280, 5120), (384, 1280, 2560), (384, 3840, 1280), (1536, 1280, 1280),
             (1536, 1280, 5120), (1536, 3840, 1280), (1536, 1280, 2560)]
    for M, N, K in gemms:
        a = torch.randn(M, K, device=dev).to(bf16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf16)
        bias = torch.randn(N, device=dev)
        line = f"gemm M{M} N{N} K{K}:"
        for s in (1, 0, 2, 4, 8, 16):
            t = timeit(lambda: ops.gemm(a, w, bias=bias, splits=s, tile=(2 if s != 1 else 0)))
            line += f"  s{s} {t * 1e6:6.1f}us"
        print(line, flush=True)
    convs = [(24, 4, 4, 1280, 1280, 1), (24, 4, 4, 2560, 1280, 1), (24, 8, 8, 1280, 1280, 2), (24, 8, 8, 1280, 1280, 1),
             (24, 8, 8, 2560, 1280, 1), (24, 8, 8, 1920, 1280, 1)]
    for n, H, W, Ci, Co, st in convs:
        x = torch.randn(n * H * W, Ci, device=dev).to(bf16)
        w = (torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(bf16)
        bias = torch.randn(Co, device=dev)
        line = f"conv n{n} {H}x{W} {Ci}->{Co} s{st}:"
        for s in (1, 0, 2, 4, 8, 16):
            t = timeit(lambda: ops.conv3x3(x, w, n, H, W, stride=st, bias=bias, splits=s, tile=(2 if s != 1 else 0)))
            line += f"  s{s} {t * 1e6:6.1f}us"
        print(line, flush=True)


def bench_split2():
    """split-K x tile sweep on the M = 1536 shapes (8x8 level): 64x64 vs 128x128 tiles under K slicing"""
    convs = [(24, 8, 8, 1280, 1280, 1), (24, 8, 8, 2560, 1280, 1), (24, 8, 8, 1920, 1280, 1), (24, 8, 8, 640, 1280, 1)]
    for n, H, W, Ci, Co, st in convs:
        x = torch.randn(n * H * W, Ci, device=dev).to(bf16)
        w = (torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(bf16)
        bias = torch.randn(Co, device=dev)
        line = f"conv n{n} {H}x{W} {Ci}->{Co}:"
        t = timeit(lambda: ops.conv3x3(x, w, n, H, W, stride=st, bias=bias))
        line += f"  auto {t * 1e6:6.1f}us |"
        for tile, name in ((8, "g64"), (5, "g128")):
            for s in (1, 2, 4, 6, 8):
                t = timeit(lambda: ops.conv3x3(x, w, n, H, W, stride=st, bias=bias, splits=s, tile=tile))
                line += f" {name}/s{s} {t * 1e6:6.1f}"
            line += " |"
        print(line, flush=True)
    gemms = [(1536, 1280, 5120), (1536, 1280, 2560), (1536, 1280, 1280)]
    for M, N, K in gemms:
        a = torch.randn(M, K, device=dev).to(bf16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf16)
        bias = torch.randn(N, device=dev)
        line = f"gemm M{M} N{N} K{K}:"
        t = timeit(lambda: ops.gemm(a, w, bias=bias))
        line += f"  auto {t * 1e6:6.1f}us |"
        for tile, name in ((8, "g64"), (5, "g128")):
            for s in (1, 2, 4, 8):
                t = timeit(lambda: ops.gemm(a, w, bias=bias, splits=s, tile=tile))
                line += f" {name}/s{s} {t * 1e6:6.1f}"
            line += " |"
        print(line, flush=True)


def bench_attn():
    cases = [  # name, batch, S_q, S_k, d, causal, window
        ("spatial L0", 24, 1024, 1024, 40, False, None), ("spatial L1", 24, 256, 256, 80, False, None),
        ("spatial L2", 24, 64, 64, 160, False, None), ("cross L0", 24, 1024, 77, 40, False, None),
        ("cross L1", 24, 256, 77, 80, False, None),
        ("temporal L0", 2, 768, 768, 40, True, (8, 12, 32, 32)), ("temporal L1", 2, 192, 192, 80, True, (4, 12, 16, 16)),
        ("temporal L2", 2, 192, 192, 160, True, (4, 12, 8, 8)),
        ("spatial 64^2", 24, 4096, 4096, 40, False, None),
    ]
    for name, B, Sq, Sk, d, causal, win in cases:
        C = 8 * d
        tq = Sq if win is None else win[1] * win[2] * win[3]
        tk = Sk if win is None else tq
        q = torch.randn(B * tq, C, device=dev).to(bf16)
        k = torch.randn(B * tk, C, device=dev).to(bf16)
        v = torch.randn(B * tk, C, device=dev).to(bf16)
        o = torch.empty(B * tq, C, device=dev, dtype=bf16)
        t = timeit(lambda: ops.attention(q, k, v, o, batch=B, heads=8, head_dim=d, Sq=Sq, Sk=Sk, causal=causal,
                                         window=win))
        nb = B if win is None else B * (win[2] // win[0]) * (win[3] // win[0])
        flops = 4 * nb * 8 * Sq * Sk * d
        print(f"attn {name}: {flops / t / 1e12:7.1f} TF dense-equivalent ({t * 1e6:7.1f} us)", flush=True)


def bench_norm():
    for rows, C in ((24576, 320), (6144, 640), (1536, 1280)):
        x = torch.randn(rows, C, device=dev).to(bf16)
        g = torch.ones(C, device=dev)
        b = torch.zeros(C, device=dev)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.layernorm(x, g, b, out=y))
        print(f"layernorm {rows}x{C}: {2 * rows * C * 2 / t / 1e9:7.1f} GB/s ({t * 1e6:6.1f} us)")
        stats = torch.zeros(2, 32, 2, device=dev)
        t = timeit(lambda: ops.groupnorm_stats(x, None, 2, 32, stats))
        print(f"gn stats  {rows}x{C}: {rows * C * 2 / t / 1e9:7.1f} GB/s ({t * 1e6:6.1f} us)")
        t = timeit(lambda: ops.groupnorm_apply(x, None, 2, 32, stats, rows // 2 * C // 32, 1e-5, g, b, True, out=y))
        print(f"gn apply  {rows}x{C}: {2 * rows * C * 2 / t / 1e9:7.1f} GB/s ({t * 1e6:6.1f} us)", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    t0 = time.time()
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("conv", "all"):
        bench_conv()
    if what in ("split", "all"):
        bench_split()
    if what in ("split2",):
        bench_split2()
    if what in ("attn", "all"):
        bench_attn()
    if what in ("norm", "all"):
        bench_norm()
    print(f"done in {time.time() - t0:.1f}s")
