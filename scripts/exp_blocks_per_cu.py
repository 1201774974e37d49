"""Experiment: how many blocks of a kernel with 64 KiB (resp. 48 / 32 KiB) of dynamic LDS are co-resident on one CU?
256 blocks = one per CU; 512 / 768 blocks finish in about the same time only if they run side by side.

    python scripts/exp_blocks_per_cu.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16


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


K = 4096
TM = 16         # M tiles: every N tile's weights are reused by 16 blocks, every M tile's rows by nblk / 16 (operands stay in L2 / MALL)
for tile, name, bm, bn in ((5, "g128x128/2 (64 KiB LDS)", 128, 128), (7, "g128x64/3 (72 KiB)", 128, 64), (8, "g64x64/3 (48 KiB)", 64, 64),
                           (2, "64x64 reg-staged (32 KiB)", 64, 64)):
    line = f"{name:28s}"
    a = torch.randn(bm * TM, K, device=dev).to(bf16)
    for nblk in (128, 256, 512, 768, 1024, 1536):
        w = (torch.randn(bn * nblk // TM, K, device=dev) * K ** -0.5).to(bf16)
        out = torch.empty(bm * TM, bn * nblk // TM, device=dev, dtype=bf16)
        t = timeit(lambda: ops.gemm(a, w, out=out, tile=tile, splits=1))
        line += f"  {nblk:4d} blocks {t:7.1f} us"
    print(line, flush=True)
