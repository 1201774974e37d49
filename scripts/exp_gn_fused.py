"""GroupNorm: the two-launch form (stats from column sums, then apply) against the fused launch, per level of the step (round 4)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def timeit(fn, n=50):
    """us per launch inside a hipGraph of n launches (what the step pays: kernel + the dependent-node boundary), not the
    host-bound eager rate"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(4):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (4 * n) * 1e3


B, G = 2, 32
print(f"{'rows/batch':>10s} {'C1':>5s} {'C2':>5s} | {'finalize us':>11s} {'apply us':>9s} {'sum':>7s} | {'fused us':>8s}")
SHAPES = [(3072, 640, 0), (3072, 640, 0), (3072, 320, 0), (1536, 640, 0), (12288, 320, 0), (12288, 320, 320), (12288, 640, 320), (3072, 640, 0), (3072, 640, 640), (3072, 1280, 640),
          (768, 1280, 0), (768, 1280, 1280), (192, 1280, 0), (192, 1280, 1280),
          (65536, 320, 0), (65536, 320, 320), (65536, 640, 320), (16384, 640, 640), (16384, 1280, 640)]       # ... and the bridge workload's (batch 2 of its 8)
for rows, C1, C2 in SHAPES:
    M = B * rows

    def produce(C, seed):
        g = torch.Generator().manual_seed(seed)
        a = torch.randn((M, 320), generator=g).to(dev).to(bf16)
        w = (torch.randn((C, 320), generator=g) * 320 ** -0.5).to(dev).to(bf16)
        y = ops.gemm(a, w, colsum_batch=B)
        assert y.colsums is not None
        return y
    x1 = produce(C1, 1)
    x2 = produce(C2, 2) if C2 else None
    C = C1 + C2
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.zeros((B, G, 2), device=dev)
    out = torch.empty((M, C), device=dev, dtype=bf16)
    cs2 = x2.colsums if C2 else None
    count = rows * (C // G)
    t_fin = timeit(lambda: ops.groupnorm_stats_from_colsums(x1.colsums, cs2, B, G, stats))
    t_app = timeit(lambda: ops.groupnorm_apply(x1, x2, B, G, stats, count, 1e-5, gamma, beta, True, out=out))
    t_fus = timeit(lambda: ops.groupnorm_apply_from_colsums(x1, x2, x1.colsums, cs2, B, G, count, 1e-5, gamma, beta, True, out=out)) \
        if ops.groupnorm_apply_from_colsums(x1, x2, x1.colsums, cs2, B, G, count, 1e-5, gamma, beta, True, out=out) is not None else 0.0
    # ... and the accumulated fixed-point form (round 4, second half): one launch, statistics straight from the int64 sums
    arena = ops.FxArena(dev, 1 << 20)
    arena.reset()

    def produce_fx(C, seed):
        g = torch.Generator().manual_seed(seed)
        a = torch.randn((M, 320), generator=g).to(dev).to(bf16)
        w = (torch.randn((C, 320), generator=g) * 320 ** -0.5).to(dev).to(bf16)
        return ops.gemm(a, w, colsum_batch=(B, arena))
    y1 = produce_fx(C1, 1)
    y2 = produce_fx(C2, 2) if C2 else None
    t_fx = timeit(lambda: ops.groupnorm_apply_fx(y1, y2, y1.colsums, y2.colsums if C2 else None, B, G, count, 1e-5, gamma, beta, True, out=out)) \
        if isinstance(y1.colsums, ops.ColSumsFx) else float("nan")
    print(f"{rows:10d} {C1:5d} {C2:5d} | {t_fin:11.2f} {t_app:9.2f} {t_fin + t_app:7.2f} | {t_fus if t_fus else float('nan'):8.2f} | fx {t_fx:7.2f}  reps {y1.colsums.reps if isinstance(y1.colsums, ops.ColSumsFx) else 0}   (partials per batch: {x1.colsums.tiles // B})")
