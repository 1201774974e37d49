"""Experiment: cost of ONE dependent kernel inside a replayed hipGraph (no profiler), for a trivial kernel and for small
GEMMs.  Sets the value of fusing kernels away.     python scripts/exp_launch_floor.py
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def graph_time(fn, n_nodes, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_nodes):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n_nodes * 1e6


x = torch.rand(64, device=dev)
print(f"clamp01 on 64 floats            : {graph_time(lambda: ops.clamp01_(x), 500):6.2f} us per node")
xb = torch.rand(24576 * 320, device=dev)
print(f"clamp01 on 31 MB (fp32, r+w)    : {graph_time(lambda: ops.clamp01_(xb), 200):6.2f} us per node")
for (M, N, K) in [(128, 128, 64), (1536, 1280, 64), (24576, 320, 64), (24576, 320, 320), (6144, 640, 640), (1536, 1280, 1280),
                  (384, 1280, 1280)]:
    a = torch.randn(M, K, device=dev).to(bf16)
    w = torch.randn(N, K, device=dev).to(bf16)
    o = torch.empty(M, N, device=dev, dtype=bf16)
    print(f"gemm M{M} N{N} K{K}".ljust(32) + f": {graph_time(lambda: ops.gemm(a, w, out=o), 200):6.2f} us per node")
h = torch.randn(24576, 320, device=dev).to(bf16)
g_ = torch.ones(320, device=dev)
b_ = torch.zeros(320, device=dev)
print(f"layernorm 24576x320             : {graph_time(lambda: ops.layernorm(h, g_, b_), 200):6.2f} us per node")
h2 = torch.randn(384, 1280, device=dev).to(bf16)
g2 = torch.ones(1280, device=dev)
b2 = torch.zeros(1280, device=dev)
print(f"layernorm 384x1280              : {graph_time(lambda: ops.layernorm(h2, g2, b2), 200):6.2f} us per node")
