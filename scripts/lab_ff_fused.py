"""The fused feed-forward launch (seer_ff_fused_c320) against the three launches it replaces, at the level-0 shape of config 2
(24 576 rows x 320 channels): microseconds per call over 50 calls, HIP events.

    python scripts/lab_ff_fused.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
M, C, inner = 24576, 320, 1280
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(s, generator=g) * sc).to(dev)
h, x = r(M, C).to(bf16), r(M, C).to(bf16)
gamma, beta = 1 + 0.1 * r(C), 0.1 * r(C)
w1, b1 = r(2 * inner, C, sc=C ** -0.5).to(bf16), 0.1 * r(2 * inner)
wcat, bcat = r(C, C + inner, sc=(C + inner) ** -0.5).to(bf16), 0.1 * r(C)
arena = ops.FxArena(dev, 1 << 20)
w1f, wcf = ops.ff_fused_pack(w1, wcat)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def unfused():
    arena.reset()
    n = ops.layernorm(h, gamma, beta)
    gg = ops.gemm(n, w1, bias=b1, geglu=True)
    return ops.gemm(h, wcat, a2=gg, bias=bcat, residual=x, colsum_batch=(2, arena))


def fused():
    arena.reset()
    return ops.ff_fused(h, x, gamma, beta, w1f, b1, wcf, bcat, colsum_batch=(2, arena))


def fused_nosum():
    return ops.ff_fused(h, x, gamma, beta, w1f, b1, wcf, bcat)


flop = 2.0 * M * C * 2 * inner + 2.0 * M * (C + inner) * C
for name, fn in (("layernorm + GEGLU projection + [proj_out | proj_out ff.net.2] GEMM", unfused), ("seer_ff_fused_c320", fused),
                 ("seer_ff_fused_c320 without column sums", fused_nosum)):
    us = timed(fn)
    print(f"{name:70s} {us:8.2f} us   {flop / us * 1e-6:7.1f} TFLOP/s   ({flop / us * 1e-6 / 2500:.3f} of the dense bf16 peak)")
ya, yb = unfused().float(), fused().float()
print("rel diff fused vs unfused:", ((ya - yb).norm() / ya.norm()).item())
# where the fused launch wins: a workgroup owns 96 rows for the whole launch, so its time steps with ceil(rows / 96 / 256 CUs)
print("rows     workgroups   three launches us   fused us")
for Mr in (6144, 12288, 18432, 24576, 36864, 49152, 98304):
    h, x = r(Mr, C).to(bf16), r(Mr, C).to(bf16)
    print(f"{Mr:6d}   {Mr // 96:6d}       {timed(unfused, 20):10.2f}      {timed(fused, 20):8.2f}")
