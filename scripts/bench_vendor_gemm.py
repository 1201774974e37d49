"""Calibration only (never on the product path): the step's plain GEMM shapes on torch.nn.functional.linear (rocBLAS /
hipBLASLt behind torch) next to seer_gemm_bf16 with the epilogue each site really uses.  A vendor GEMM has no GEGLU /
residual / rotary epilogue, so its column is the bare product (+ bias): what it would still have to add is listed.

    python scripts/bench_vendor_gemm.py > gpurun_out/vendor_gemm.log
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3      # us


SHAPES = [  # M, N, K, what
    (24576, 2560, 320, "geglu"), (6144, 5120, 640, "geglu"), (1536, 10240, 1280, "geglu"),
    (24576, 320, 1280, "res"), (6144, 640, 2560, "res"), (1536, 1280, 5120, "res"),
    (24576, 960, 320, "plain"), (6144, 1920, 640, "plain"), (1536, 3840, 1280, "plain"),
    (24576, 320, 320, "res"), (6144, 640, 640, "res"), (1536, 1280, 1280, "res"), (384, 1280, 1280, "res"),
    (4096, 4096, 4096, "plain"), (8192, 8192, 8192, "plain"),
    # N a multiple of 320: AUTO takes the 256 x 320 tile kernel (gemm_t320.hip) where its cost model puts it ahead
    (4096, 3840, 4096, "plain"), (8192, 7680, 8192, "plain"), (16384, 5120, 2560, "plain"),
    (24576, 5120, 640, "geglu"), (6144, 10240, 1280, "geglu"), (98304, 320, 1280, "res"), (24576, 640, 2560, "res"),
]

print(f"{'shape':28s} {'epilogue':8s} {'ours us':>9s} {'TF/s':>7s} | {'torch linear us':>15s} {'TF/s':>7s} | {'+ what torch still owes':s}")
for M, N, K, what in SHAPES:
    a = torch.randn(M, K, device=dev).to(bf16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf16)
    bias = torch.randn(N, device=dev)
    bias_b = bias.to(bf16)
    fl = 2.0 * M * N * K
    if what == "geglu":
        out = torch.empty((M, N // 2), device=dev, dtype=bf16)
        t_our = timeit(lambda: ops.gemm(a, w, bias=bias, geglu=True, out=out))
        owes = "chunk + gelu + mul pass over [M, N]"
    elif what == "res":
        res = torch.randn(M, N, device=dev).to(bf16)
        out = torch.empty((M, N), device=dev, dtype=bf16)
        t_our = timeit(lambda: ops.gemm(a, w, bias=bias, residual=res, out=out))
        owes = "residual add pass over [M, N]"
    else:
        out = torch.empty((M, N), device=dev, dtype=bf16)
        t_our = timeit(lambda: ops.gemm(a, w, out=out))
        owes = "-"
    t_ref = timeit(lambda: F.linear(a, w, bias_b if what != "plain" else None))
    print(f"M{M} N{N} K{K}".ljust(28) + f" {what:8s} {t_our:9.1f} {fl / t_our / 1e6:7.0f} | {t_ref:15.1f} {fl / t_ref / 1e6:7.0f} | {owes}",
          flush=True)
