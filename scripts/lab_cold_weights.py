"""3x3 convs and GEMMs of the lower levels with COLD weights: inside the step every launch reads weights that were last touched a step
(1.7 GB of other weights) ago, i.e. from HBM; a back-to-back lab over ONE weight tensor reads them from the 256 MB memory-side cache.
Here each shape cycles through enough weight tensors to exceed it (>= 400 MB), inside a replayed hipGraph, and the tile / K-split
choices are compared under that condition.

    python scripts/lab_cold_weights.py > profiles/r06_lab_cold_weights.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def timed(fns, replays=4):
    """fns: the launches of one graph (one per weight tensor); us per launch"""
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


def conv_case(name, n_img, H, Ci, Co, variants):
    wbytes = Co * 9 * Ci * 2
    nw = max(2, -(-420_000_000 // wbytes))
    ws = [(torch.randn((Co, 9 * Ci), device=dev) * (9 * Ci) ** -0.5).to(bf16) for _ in range(nw)]
    x = torch.randn((n_img * H * H, Ci), device=dev).to(bf16)
    out = torch.empty((n_img * H * H, Co), device=dev, dtype=bf16)
    cells = []
    for tile, splits in variants:
        try:
            hot = timed([lambda: ops.conv3x3(x, ws[0], n_img, H, H, out=out, tile=tile, splits=splits)] * nw)
            cold = timed([(lambda w=w: ops.conv3x3(x, w, n_img, H, H, out=out, tile=tile, splits=splits)) for w in ws])
            cells.append(f"{hot:6.1f}/{cold:6.1f}")
        except Exception as e:
            cells.append(f"{'-':>13s}")
    print(f"{name:26s} {wbytes / 1e6:6.1f} MB x{nw:3d} | " + " ".join(f"{c:>13s}" for c in cells), flush=True)


def gemm_case(name, M, N, K, variants):
    wbytes = N * K * 2
    nw = max(2, -(-420_000_000 // wbytes))
    ws = [(torch.randn((N, K), device=dev) * K ** -0.5).to(bf16) for _ in range(nw)]
    x = torch.randn((M, K), device=dev).to(bf16)
    out = torch.empty((M, N), device=dev, dtype=bf16)
    cells = []
    for tile, splits in variants:
        try:
            hot = timed([lambda: ops.gemm(x, ws[0], out=out, tile=tile, splits=splits)] * nw)
            cold = timed([(lambda w=w: ops.gemm(x, w, out=out, tile=tile, splits=splits)) for w in ws])
            cells.append(f"{hot:6.1f}/{cold:6.1f}")
        except Exception:
            cells.append(f"{'-':>13s}")
    print(f"{name:26s} {wbytes / 1e6:6.1f} MB x{nw:3d} | " + " ".join(f"{c:>13s}" for c in cells), flush=True)


CV = [(0, 0), (5, 4), (5, 8), (5, 16), (5, 24), (16, 8), (16, 16), (8, 8), (8, 16), (8, 32)]
print("us per launch, hot / cold weights; columns: (tile, splits) = " + " ".join(f"{str(v):>13s}" for v in CV))
print("tiles: 0 auto, 5 128x128/2 stages, 16 96x160/2, 8 64x64/3")
conv_case("conv 4x4 1280->1280", 24, 4, 1280, 1280, CV)
conv_case("conv 4x4 2560->1280", 24, 4, 2560, 1280, CV)
conv_case("conv 8x8 1280->1280", 24, 8, 1280, 1280, CV)
conv_case("conv 8x8 2560->1280", 24, 8, 2560, 1280, CV)
conv_case("conv 16x16 640->640", 24, 16, 640, 640, CV)
conv_case("conv 16x16 1280->640", 24, 16, 1280, 640, CV)
GV = [(0, 0), (8, 1), (9, 1), (10, 1), (7, 1), (11, 1), (5, 1), (5, 2), (8, 2), (8, 4)]
print("\ncolumns: (tile, splits) = " + " ".join(f"{str(v):>13s}" for v in GV))
print("tiles: 8 / 9 / 10 64x64 with 3 / 4 / 5 stages, 7 / 11 128x64 with 3 / 4 stages, 5 128x128/2")
gemm_case("L2 proj 1536x1280x1280", 1536, 1280, 1280, GV)
gemm_case("L2 qkv 1536x3840x1280", 1536, 3840, 1280, GV)
gemm_case("L2 ff1 1536x10240x1280", 1536, 10240, 1280, GV)
gemm_case("L2 ff2 1536x1280x6400", 1536, 1280, 6400, GV)
gemm_case("L1 proj 6144x640x640", 6144, 640, 640, GV)
gemm_case("L1 ff1 6144x5120x640", 6144, 5120, 640, GV)
gemm_case("L1 ff2 6144x640x3200", 6144, 640, 3200, GV)
gemm_case("mid proj 384x1280x1280", 384, 1280, 1280, GV)
