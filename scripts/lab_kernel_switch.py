"""Does a launch cost more when the CU just ran ANOTHER kernel?  Pairs of step launches timed in a replayed graph back to back (A x n, B x n)
and interleaved (A B A B ...): interleaved minus the sum of the two = what switching kernels costs (instruction fetch, LDS / register
re-partitioning), with operands hot in both cases.

    python scripts/lab_kernel_switch.py > profiles/r06_lab_kernel_switch.log
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


def timed(fns, replays=5):
    for f in fns[:4]:
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


x = r((M, C), 1.5).to(bf16)
gg, gb, lg, lb = r((C,)) * 0.2 + 1, r((C,)) * 0.2, r((C,)) * 0.2 + 1, r((C,)) * 0.2
bp = r((C,)) * 0.1
wpf = ops.rowchain_pack(r((C, C), C ** -0.5).to(bf16))
wqkvf = ops.rowchain_pack(r((3 * C, C), C ** -0.5).to(bf16))
stats = torch.zeros((B, G, 2), device=dev)
ops.groupnorm_stats(x, None, B, G, stats)
count = rows_pb * (C // G)
sc = ops.qk_prescale(40)
h_o, qkv_o = torch.empty((M, C), device=dev, dtype=bf16), torch.empty((M, 3 * C), device=dev, dtype=bf16)
wconv = (r((C, 9 * C)) * (9 * C) ** -0.5).to(bf16)
conv_o = torch.empty((M, C), device=dev, dtype=bf16)
q = r((M, 3 * C)).to(bf16)
att_o = torch.empty((M, C), device=dev, dtype=bf16)
xa_o = torch.empty((M, C), device=dev, dtype=bf16)
w1 = (r((1280, 1280)) * 1280 ** -0.5).to(bf16)
x2 = r((1536, 1280)).to(bf16)
o2 = torch.empty((1536, 1280), device=dev, dtype=bf16)

K = {
    "rowchain F1": lambda: ops.rowchain(x, wpf, b1=bp, gn=(stats, count, 1e-6, gg, gb, rows_pb), ln=(lg, lb, 1e-5), w2f=wqkvf, col_scale=(sc, 1),
                                        h_out=h_o, out=qkv_o),
    "conv3x3 32x32 320->320": lambda: ops.conv3x3(x, wconv, 24, 32, 32, out=conv_o),
    "attention spatial d40": lambda: ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], att_o, batch=24, heads=8, head_dim=40, Sq=1024, Sk=1024),
    "groupnorm apply": lambda: ops.groupnorm_apply(x, None, B, G, stats, count, 1e-6, gg, gb, True, out=xa_o),
    "proj 1536x1280x1280": lambda: ops.gemm(x2, w1, out=o2),
}
n = 12
alone = {k: timed([f] * (2 * n)) for k, f in K.items()}
for k, v in alone.items():
    print(f"{k:28s} back to back {v:7.1f} us")
names = list(K)
print()
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = names[i], names[j]
        t = timed([K[a], K[b]] * n) * 2          # us per (A, B) pair
        print(f"{a:28s} + {b:28s} interleaved {t:7.1f} us per pair, back to back {alone[a] + alone[b]:7.1f}  ({t - alone[a] - alone[b]:+5.1f})", flush=True)
