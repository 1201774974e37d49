"""split-K x tile sweep on the 4x4-level (M = 384) shapes."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from scripts.bench_kernels import timeit, ops, bf16, dev  # noqa: E402

for n, H, W, Ci, Co, st in [(24, 4, 4, 1280, 1280, 1), (24, 4, 4, 2560, 1280, 1), (24, 8, 8, 1280, 1280, 2)]:
    x = torch.randn(n * H * W, Ci, device=dev).to(bf16)
    w = (torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(bf16)
    bias = torch.randn(Co, device=dev)
    line = f"conv n{n} {H}x{W} {Ci}->{Co} s{st}: auto {timeit(lambda: ops.conv3x3(x, w, n, H, W, stride=st, bias=bias)) * 1e6:6.1f} |"
    for tile, name in ((8, "g64"), (5, "g128")):
        for s in (4, 6, 8, 12, 16):
            t = timeit(lambda: ops.conv3x3(x, w, n, H, W, stride=st, bias=bias, splits=s, tile=tile))
            line += f" {name}/s{s} {t * 1e6:6.1f}"
        line += " |"
    print(line, flush=True)
for M, N, K in [(384, 1280, 5120), (384, 1280, 2560), (384, 1280, 1280), (384, 3840, 1280)]:
    a = torch.randn(M, K, device=dev).to(bf16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf16)
    line = f"gemm M{M} N{N} K{K}: auto {timeit(lambda: ops.gemm(a, w)) * 1e6:6.1f} |"
    for tile, name in ((8, "g64"), (5, "g128")):
        for s in (1, 2, 4, 8):
            t = timeit(lambda: ops.gemm(a, w, splits=s, tile=tile))
            line += f" {name}/s{s} {t * 1e6:6.1f}"
        line += " |"
    print(line, flush=True)
