"""Tile x split-K sweep of the step's GEMM / conv shapes at a given (batch, frames): the per-rank shapes of a sharded step
(B = 1 at 2 GPUs, fewer frames per rank at 4 / 8) differ from the single-GPU ones the heuristics were tuned on.

    python scripts/sweep_shapes.py [B=1] [frames=12]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Fr = int(sys.argv[2]) if len(sys.argv) > 2 else 12
TILES = ((8, "g64/3"), (7, "g128x64/3"), (5, "g128/2"), (12, "g128x160/2"))
SPLITS = (1, 2, 4, 8, 16)


def timeit(fn, iters=15, warmup=3):
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


def sweep(name, run):
    timeit(lambda: run(0, 0), iters=60)        # clocks / caches settle first: the first series of a shape reads ~10 % slow
    t_auto = timeit(lambda: run(0, 0))
    best = (t_auto, "auto")
    cells = []
    for tile, tname in TILES:
        for s in SPLITS:
            try:
                t = timeit(lambda: run(tile, s))
            except Exception:
                continue
            cells.append((t, f"{tname}/s{s}"))
            if t < best[0]:
                best = (t, f"{tname}/s{s}")
    cells.sort()
    top = "  ".join(f"{n} {t:.1f}" for t, n in cells[:3])
    flag = "  <-- auto is >8% off" if best[0] < 0.92 * t_auto else ""
    print(f"{name:44s} auto {t_auto:7.1f} us | best {best[1]:16s} {best[0]:7.1f} us | {top}{flag}", flush=True)


n_img = B * Fr
levels = [(320, 32), (640, 16), (1280, 8), (1280, 4)]
print(f"# B={B} frames={Fr} (n_img={n_img})")
for C, H in levels:
    M = n_img * H * H
    # transformer GEMMs of the level (the 4x4 level has none)
    if H > 4:
        for (N, K, what) in ((3 * C, C, "qkv"), (C, C, "proj+res"), (C, 4 * C, "ff2+res")):
            a = torch.randn(M, K, device=dev).to(bf16)
            w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf16)
            bias = torch.randn(N, device=dev)
            res = torch.randn(M, N, device=dev).to(bf16) if "res" in what else None
            out = torch.empty(M, N, device=dev, dtype=bf16)
            sweep(f"gemm M{M} N{N} K{K} {what}", lambda t, s: ops.gemm(a, w, bias=bias, residual=res, out=out, tile=t, splits=s))
    # resnet convs
    for Ci in sorted({C, 2 * C if C < 1280 else 2560}):
        x = torch.randn(n_img * H * H, Ci, device=dev).to(bf16)
        w = (torch.randn(C, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(bf16)
        bias = torch.randn(C, device=dev)
        sweep(f"conv n{n_img} {H}x{H} {Ci}->{C}", lambda t, s: ops.conv3x3(x, w, n_img, H, H, bias=bias, tile=t, splits=s))
    # GEGLU feed-forward projection, 1x1 shortcut of the up blocks (K = concat channels), down / up-sample convs
    if H > 4:
        a = torch.randn(M, C, device=dev).to(bf16)
        w = (torch.randn(8 * C, C, device=dev) * C ** -0.5).to(bf16)
        bias = torch.randn(8 * C, device=dev)
        out = torch.empty(M, 4 * C, device=dev, dtype=bf16)
        sweep(f"gemm M{M} N{8 * C} K{C} geglu", lambda t, s: ops.gemm(a, w, bias=bias, geglu=True, out=out, tile=t, splits=1) if s == 1 or t == 0 else (_ for _ in ()).throw(RuntimeError()))
    for Ci in sorted({2 * C, 3 * C} if C < 1280 else {2560}):
        a = torch.randn(M, Ci, device=dev).to(bf16)
        w = (torch.randn(C, Ci, device=dev) * Ci ** -0.5).to(bf16)
        bias = torch.randn(C, device=dev)
        out = torch.empty(M, C, device=dev, dtype=bf16)
        sweep(f"gemm M{M} N{C} K{Ci} shortcut", lambda t, s: ops.gemm(a, w, bias=bias, out=out, tile=t, splits=s))
    if H > 4:
        x = torch.randn(n_img * H * H, C, device=dev).to(bf16)
        w = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).to(bf16)
        bias = torch.randn(C, device=dev)
        sweep(f"conv n{n_img} {H}x{H} {C}->{C} stride 2", lambda t, s: ops.conv3x3(x, w, n_img, H, H, stride=2, bias=bias, tile=t, splits=s))
    if H < 32:
        x = torch.randn(n_img * H * H, C, device=dev).to(bf16)
        w = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).to(bf16)
        bias = torch.randn(C, device=dev)
        sweep(f"conv n{n_img} {H}x{H} {C}->{C} upsample", lambda t, s: ops.conv3x3(x, w, n_img, H, H, upsample=True, bias=bias, tile=t, splits=s))
