"""the fused feed-forward with the attention's to_out + residual as its prologue (seer_ff_fused_c320_pre) against the two launches it
replaces, back to back inside a replayed hipGraph, at 24 576 rows.

    python scripts/lab_ff_pre.py > profiles/r06_lab_ff_pre.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402
from seervideoldm_amd.weights import geglu_row_order  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
C, inner, M = 320, 1280, 24576


def r(shape, s=1.0):
    return torch.randn(shape, device=dev) * s


def timed(fn, per_graph=20, replays=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (per_graph * replays) * 1e3


a, h, x = r((M, C)).to(bf16), r((M, C)).to(bf16), r((M, C)).to(bf16)
wo, bo = r((C, C), C ** -0.5).to(bf16), r((C,)) * 0.1
gamma, beta = r((C,)) * 0.2 + 1, r((C,)) * 0.1
order = geglu_row_order(inner, dev)
w1 = r((2 * inner, C), C ** -0.5).to(bf16)[order].contiguous()
b1 = (r((2 * inner,)) * 0.2)[order].contiguous()
wcat, bcat = r((C, C + inner), (C + inner) ** -0.5).to(bf16), r((C,)) * 0.2
w1f, wcf = ops.ff_fused_pack(w1, wcat)
wof = ops.rowchain_pack(wo)
h2 = torch.empty_like(h)
t2 = timed(lambda: ops.ff_fused(ops.gemm(a, wo, bias=bo, residual=h, out=h2), x, gamma, beta, w1f, b1, wcf, bcat))
t1 = timed(lambda: ops.ff_fused(h, x, gamma, beta, w1f, b1, wcf, bcat, pre=(a, wof, bo)))
t0 = timed(lambda: ops.ff_fused(h, x, gamma, beta, w1f, b1, wcf, bcat))
print(f"{M} rows: to_out + residual, then the fused feed-forward: {t2:.1f} us; one launch with the prologue: {t1:.1f} us (the fused feed-forward alone: {t0:.1f} us)")
