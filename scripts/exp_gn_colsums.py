"""GroupNorm statistics: the two-stage pass over the activations (seer_groupnorm_stats) against the column-sum form
(seer_groupnorm_stats_from_colsums) on the engine's shapes, each inside a replayed HIP graph of 20 launches (what a launch
costs inside the captured step).  The producer's extra work for the column sums is measured separately by
`LAB_COLSUM=1 build/lab_gemm`.

    python scripts/exp_gn_colsums.py > gpurun_out/gn_colsums.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

bf16 = torch.bfloat16


def graph_us(fn, reps=20, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * iters)


def main():
    dev = torch.device("cuda:0")
    B, G = 2, 32
    print("B x rows x (C1+C2), partial rows per source      two-stage us   from colsums us")
    # (rows per batch element, C1, C2, rows per partial of source 1, of source 2)
    for rows, C1, C2, r1, r2 in [(12288, 320, 0, 96, 0), (12288, 320, 0, 64, 0), (12288, 640, 320, 96, 64),
                                 (3072, 640, 0, 16, 0), (3072, 640, 0, 128, 0), (3072, 1280, 640, 16, 16),
                                 (768, 1280, 0, 16, 0), (768, 1280, 1280, 16, 16), (192, 1280, 0, 16, 0),
                                 (192, 1280, 1280, 16, 16)]:
        x1 = torch.randn((B * rows, C1), device=dev).to(bf16)
        x2 = torch.randn((B * rows, C2), device=dev).to(bf16) if C2 else None
        stats = torch.zeros((B, G, 2), device=dev)
        cs1 = ops.ColSums(torch.randn((1, B * rows // r1, C1, 2), device=dev), C1, 1, B * rows // r1)
        cs2 = ops.ColSums(torch.randn((1, B * rows // r2, C2, 2), device=dev), C2, 1, B * rows // r2) if C2 else None
        t_two = graph_us(lambda: ops.groupnorm_stats(x1, x2, B, G, stats))
        t_cs = graph_us(lambda: ops.groupnorm_stats_from_colsums(cs1, cs2, B, G, stats))
        print(f"{B} x {rows} x ({C1}+{C2}), {B * rows // r1}" + (f" / {B * rows // r2}" if C2 else "") +
              f"   {t_two:10.2f}   {t_cs:10.2f}")


if __name__ == "__main__":
    main()
