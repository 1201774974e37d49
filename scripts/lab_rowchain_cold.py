"""Why seer_rowchain_c320 takes 50 us inside the step and 35-40 us back to back: the same launch with (a) everything hot -- one input, one
set of weights, outputs rewritten in place; (b) another set of weights per launch (>= 400 MB in rotation); (c) another INPUT per launch
(activations that a previous kernel wrote and nobody has read yet); (d) other OUTPUT buffers per launch; (e) all three.

    python scripts/lab_rowchain_cold.py > profiles/r06_lab_rowchain_cold.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
C, G = 320, 32
B, rows_pb = 2, 12288
M = B * rows_pb


def r(shape, s=1.0):
    return torch.randn(shape, device=dev) * s


def timed(fns, replays=4):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (len(fns) * replays) * 1e3


N = 24                                   # launches per graph: 24 inputs x 15.7 MB = 377 MB, 24 x 63 MB of outputs
xs = [r((M, C), 1.5).to(bf16) for _ in range(N)]
gg, gb, lg, lb = r((C,)) * 0.2 + 1, r((C,)) * 0.2, r((C,)) * 0.2 + 1, r((C,)) * 0.2
bp = r((C,)) * 0.1
NW = 520                                 # weight sets: 520 x 0.8 MB = 426 MB
wpfs = [ops.rowchain_pack(r((C, C), C ** -0.5).to(bf16)) for _ in range(NW)]
wqkvfs = [ops.rowchain_pack(r((3 * C, C), C ** -0.5).to(bf16)) for _ in range(NW)]
stats = torch.zeros((B, G, 2), device=dev)
ops.groupnorm_stats(xs[0], None, B, G, stats)
count = rows_pb * (C // G)
sc = ops.qk_prescale(40)
hs = [torch.empty((M, C), device=dev, dtype=bf16) for _ in range(N)]
outs = [torch.empty((M, 3 * C), device=dev, dtype=bf16) for _ in range(N)]


def launch(i, cold_w, cold_x, cold_o):
    wi = (i * 21) % NW if cold_w else 0
    xi = i if cold_x else 0
    oi = i if cold_o else 0
    return lambda: ops.rowchain(xs[xi], wpfs[wi], b1=bp, gn=(stats, count, 1e-6, gg, gb, rows_pb), ln=(lg, lb, 1e-5), w2f=wqkvfs[wi],
                                col_scale=(sc, 1), h_out=hs[oi], out=outs[oi])


for name, cw, cx, co in (("everything hot", 0, 0, 0), ("cold weights", 1, 0, 0), ("cold input", 0, 1, 0), ("cold outputs", 0, 0, 1),
                         ("cold input + outputs", 0, 1, 1), ("all cold", 1, 1, 1)):
    if cw:      # more launches per graph so that the weight rotation exceeds the cache
        fns = [launch(i % N if (cx or co) else i, cw, cx, co) for i in range(NW)]
    else:
        fns = [launch(i, cw, cx, co) for i in range(N)] * 4
    print(f"GroupNorm -> proj_in -> norm1 -> q|k|v, 24 576 rows, {name:22s} {timed(fns):7.1f} us", flush=True)
