"""Ring depth on FEW-TILE GEMMs (grids that do not fill the 256 CUs: the b = 1 fine-tuning step, FSTextTransformer's 924-row
products, the 8x8 / 4x4 levels, per-rank shapes of a sharded step): with one workgroup per CU the K loop is a chain of L2
round trips and LDS is free, so a deeper LDS-DMA ring should shorten it.

    python scripts/sweep_small.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
TILES = ((0, "auto"), (2, "reg64"), (8, "g64/3"), (9, "g64/4"), (10, "g64/5"), (7, "g128x64/3"), (11, "g128x64/4"), (5, "g128/2"))
SHAPES = [("fstext qkv", 924, 2304, 768), ("fstext out/q", 924, 768, 768), ("fstext ff1", 924, 6144, 768), ("fstext ff2", 924, 768, 3072),
          ("fstext kv (ctx)", 77, 1536, 768),
          ("L1 proj", 3072, 640, 640), ("L1 qkv", 3072, 1920, 640), ("L1 ff1", 3072, 5120, 640), ("L1 ff2", 3072, 640, 2560),
          ("L2 proj", 768, 1280, 1280), ("L2 qkv", 768, 3840, 1280), ("L2 ff1", 768, 10240, 1280), ("L2 ff2", 768, 1280, 5120),
          ("mid proj", 192, 1280, 1280), ("mid ff1", 192, 10240, 1280), ("mid ff2", 192, 1280, 5120),
          ("L0 proj b1", 12288, 320, 320), ("L0 ff2 b1", 12288, 320, 1280)]


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print(f"{'shape':18s} {'M':>6s} {'N':>6s} {'K':>5s} | " + " ".join(f"{n:>10s}" for _, n in TILES))
for name, M, N, K in SHAPES:
    a = torch.randn((M, K), device=dev).to(bf16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(bf16)
    out = torch.empty((M, N), device=dev, dtype=bf16)
    timeit(lambda: ops.gemm(a, w, out=out), iters=40)
    cells = []
    for tile, _ in TILES:
        try:
            cells.append(timeit(lambda: ops.gemm(a, w, out=out, tile=tile, splits=1 if tile else 0)))
        except Exception:
            cells.append(float("nan"))
    best = min(c for c in cells[1:] if c == c)
    flag = "  <--" if best < 0.93 * cells[0] else ""
    print(f"{name:18s} {M:6d} {N:6d} {K:5d} | " + " ".join(f"{c:10.1f}" for c in cells) + flag, flush=True)
