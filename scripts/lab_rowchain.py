"""seer_rowchain_c320 against the launches it replaces, back to back inside a replayed hipGraph (as the denoising step runs them), at the
32x32 level of BASELINE config 2 (24 576 rows) and at a rank's shares.

    python scripts/lab_rowchain.py > profiles/r06_lab_rowchain.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
C, G = 320, 32


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


print("rows      chain                                                     launches it replaces, us        one launch, us")
for B, rows_pb in ((2, 12288), (1, 12288), (1, 6144), (1, 3072)):
    M = B * rows_pb
    x = r((M, C), 1.5).to(bf16)
    gg, gb, lg, lb = r((C,)) * 0.2 + 1, r((C,)) * 0.2, r((C,)) * 0.2 + 1, r((C,)) * 0.2
    wp, bp = r((C, C), C ** -0.5).to(bf16), r((C,)) * 0.1
    wqkv = r((3 * C, C), C ** -0.5)
    stats = torch.zeros((B, G, 2), device=dev)
    ops.groupnorm_stats(x, None, B, G, stats)
    count = rows_pb * (C // G)
    arena = ops.FxArena(dev, M * 2 * 8)
    wf, wsum, bfold = ops.fold_layernorm(wqkv, lg, lb)
    sc = ops.qk_prescale(40)
    freqs = (10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)).to(dev)
    table = ops.rotary_table(freqs, rows_pb)
    wpf, wqkvf = ops.rowchain_pack(wp), ops.rowchain_pack(wqkv.to(bf16))

    for rot in (False, True):
        def three():
            arena.reset()
            xa = ops.groupnorm_apply(x, None, B, G, stats, count, 1e-6, gg, gb, False)
            h = ops.gemm(xa, wp, bias=bp, rowstat=arena)
            return ops.gemm(h, wf, bias=bfold, ln=(h.rowstats, wsum, 1e-5), col_scale=(sc, C),
                            rotary=(table, rows_pb, 0, 40, 32, 2 * C) if rot else None)

        def one():
            return ops.rowchain(x, wpf, b1=bp, gn=(stats, count, 1e-6, gg, gb, rows_pb), ln=(lg, lb, 1e-5), w2f=wqkvf, col_scale=(sc, 1),
                                rotary=(table, rows_pb, 0, 40, 32, 2) if rot else None)
        t3, t1 = timed(three), timed(one)
        print(f"{M:6d}    GroupNorm -> proj_in -> norm1 -> q|k|v{' (rotary)' if rot else '         '}          {t3:8.1f} (incl. one 1.5 us fill)   {t1:8.1f}")
    a = r((M, C)).to(bf16)
    h0 = r((M, C)).to(bf16)
    wo, bo = r((C, C), C ** -0.5).to(bf16), r((C,)) * 0.1
    wq = r((C, C), C ** -0.5)
    wqf_, wqsum, bq = ops.fold_layernorm(wq, lg, lb)
    wof, wqf = ops.rowchain_pack(wo), ops.rowchain_pack(wq.to(bf16))

    def two():
        arena.reset()
        h = ops.gemm(a, wo, bias=bo, residual=h0, out=h0, rowstat=arena)
        return ops.gemm(h, wqf_, bias=bq, ln=(h.rowstats, wqsum, 1e-5), col_scale=(sc, C))

    def one2():
        return ops.rowchain(a, wof, b1=bo, res=h0, h_out=h0, ln=(lg, lb, 1e-5), w2f=wqf, col_scale=(sc, 1))
    t2, t1 = timed(two), timed(one2)
    print(f"{M:6d}    to_out + residual -> norm2 -> to_q                        {t2:8.1f} (incl. one 1.5 us fill)   {t1:8.1f}")
