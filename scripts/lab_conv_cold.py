"""The 3x3 conv of the 32x32 level (24 576 rows, 320 -> 320) hot, with another INPUT per launch, other OUTPUT buffers, other weights,
and all three: what the launch loses inside the step against its back-to-back time.  Same for the pair GroupNorm apply -> conv (the conv
reading what the apply just wrote), as the step runs them.

    python scripts/lab_conv_cold.py > profiles/r06_lab_conv_cold.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
C, G, B, rows_pb = 320, 32, 2, 12288
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


N = 24
xs = [r((M, C), 1.5).to(bf16) for _ in range(N)]
xas = [torch.empty((M, C), device=dev, dtype=bf16) for _ in range(N)]
outs = [torch.empty((M, C), device=dev, dtype=bf16) for _ in range(N)]
ws = [(r((C, 9 * C)) * (9 * C) ** -0.5).to(bf16) for _ in range(240)]       # 240 x 1.8 MB = 442 MB
gg, gb = r((C,)) * 0.2 + 1, r((C,)) * 0.2
stats = torch.zeros((B, G, 2), device=dev)
ops.groupnorm_stats(xs[0], None, B, G, stats)
count = rows_pb * (C // G)


def conv(i, cw, cx, co):
    return lambda: ops.conv3x3(xs[i if cx else 0], ws[(7 * i) % 240 if cw else 0], 24, 32, 32, out=outs[i if co else 0])


for name, cw, cx, co in (("everything hot", 0, 0, 0), ("cold weights", 1, 0, 0), ("cold input", 0, 1, 0), ("cold outputs", 0, 0, 1),
                         ("cold input + outputs", 0, 1, 1), ("all cold", 1, 1, 1)):
    n = 240 if cw else N * 4
    print(f"conv3x3 32x32 320 -> 320, 24 576 rows, {name:22s} {timed([conv(i % N, cw, cx, co) for i in range(n)]):7.1f} us", flush=True)


def pair(i, cold):
    j = i if cold else 0
    def f():
        ops.groupnorm_apply(xs[j], None, B, G, stats, count, 1e-6, gg, gb, True, out=xas[j])
        ops.conv3x3(xas[j], ws[(7 * i) % 240 if cold else 0], 24, 32, 32, out=outs[j])
    return f


ta = timed([lambda: ops.groupnorm_apply(xs[0], None, B, G, stats, count, 1e-6, gg, gb, True, out=xas[0])] * 96)
print(f"GroupNorm apply + SiLU alone, hot                                {ta:7.1f} us")
print(f"apply -> conv, one set of buffers                                 {timed([pair(i, 0) for i in range(96)]):7.1f} us per pair")
print(f"apply -> conv, rotating buffers and weights                       {timed([pair(i % N, 1) for i in range(240)]):7.1f} us per pair")
