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


def gemm_case(name, M, N, K, variants, geglu=False):
    wbytes = N * K * 2
    nw = max(2, -(-420_000_000 // wbytes))
    ws = [(torch.randn((N, K), device=dev) * K ** -0.5).to(bf16) for _ in range(nw)]
    x = torch.randn((M, K), device=dev).to(bf16)
    out = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=bf16)
    cells = []
    for tile, splits in variants:
        try:
            hot = timed([lambda: ops.gemm(x, ws[0], out=out, tile=tile, splits=splits, geglu=geglu)] * nw)
            cold = timed([(lambda w=w: ops.gemm(x, w, out=out, tile=tile, splits=splits, geglu=geglu)) for w in ws])
            cells.append(f"{hot:6.1f}/{cold:6.1f}")
        except Exception:
            cells.append(f"{'-':>13s}")
    print(f"{name:26s} {wbytes / 1e6:6.1f} MB x{nw:3d} | " + " ".join(f"{c:>13s}" for c in cells), flush=True)


def run(kind, name, dims, variants, **kw):
    print(f"{'':45s}" + " ".join(f"{str(v):>13s}" for v in variants))
    (conv_case if kind == "conv" else gemm_case)(name, *dims, variants, **kw)


def main():
    print("us per launch, hot / cold weights, per (tile, splits); tiles: 0 auto, 5 128x128/2 stages, 16 96x160/2, 18 96x128/2, 12 128x160/2, "
          "8 / 10 64x64 with 3 / 5 stages, 7 128x64/3, 14 256x128/2")
    C4 = [(0, 0), (5, 8), (5, 16), (16, 4), (16, 8), (16, 12), (16, 16), (18, 8), (12, 8), (14, 8)]
    run("conv", "conv 4x4 1280->1280", (24, 4, 1280, 1280), C4)
    run("conv", "conv 4x4 2560->1280", (24, 4, 2560, 1280), C4)
    C8 = [(0, 0), (5, 2), (5, 4), (16, 2), (16, 4), (18, 2), (18, 4), (12, 2), (12, 4), (14, 4)]
    run("conv", "conv 8x8 1280->1280", (24, 8, 1280, 1280), C8)
    run("conv", "conv 8x8 2560->1280", (24, 8, 2560, 1280), C8)
    run("conv", "conv 8x8 1920->1280", (24, 8, 1920, 1280), C8)
    run("conv", "conv 8x8 640->1280", (24, 8, 640, 1280), C8)
    C16 = [(0, 0), (5, 1), (5, 2), (16, 1), (16, 2), (18, 1), (18, 2), (12, 1), (12, 2), (14, 2)]
    run("conv", "conv 16x16 640->640", (24, 16, 640, 640), C16)
    run("conv", "conv 16x16 1280->640", (24, 16, 1280, 640), C16)
    run("conv", "conv 16x16 1920->640", (24, 16, 1920, 640), C16)
    run("conv", "conv 16x16 960->640", (24, 16, 960, 640), C16)
    run("conv", "conv 16x16 320->640", (24, 16, 320, 640), C16)
    GV = [(0, 0), (8, 1), (10, 1), (7, 1), (5, 1), (18, 1), (12, 1), (5, 2), (5, 4), (8, 4)]
    print()
    for nm, dims in (("L2 proj 1536x1280x1280", (1536, 1280, 1280)), ("L2 qkv 1536x3840x1280", (1536, 3840, 1280)),
                     ("L2 ff2 1536x1280x6400", (1536, 1280, 6400)), ("L2 shortcut 1536x1280x2560", (1536, 1280, 2560)),
                     ("L1 proj 6144x640x640", (6144, 640, 640)), ("L1 qkv 6144x1920x640", (6144, 1920, 640)),
                     ("L1 ff2 6144x640x3200", (6144, 640, 3200)), ("L1 shortcut 6144x640x1280", (6144, 640, 1280)),
                     ("L1 shortcut 6144x640x1920", (6144, 640, 1920)),
                     ("mid proj 384x1280x1280", (384, 1280, 1280)), ("mid qkv 384x3840x1280", (384, 3840, 1280)),
                     ("mid ff2 384x1280x6400", (384, 1280, 6400)), ("L3 shortcut 384x1280x2560", (384, 1280, 2560))):
        run("gemm", nm, dims, GV)
    GG = [(0, 0), (5, 1), (18, 1), (12, 1), (7, 1), (19, 1), (20, 1)]
    print("\nGEGLU projections (tile 19 = weight-stationary, 20 = AUTO restricted to the tile kernels)")
    for nm, dims in (("L1 ff1 geglu 6144x5120x640", (6144, 5120, 640)), ("L2 ff1 geglu 1536x10240x1280", (1536, 10240, 1280)),
                     ("mid ff1 geglu 384x10240x1280", (384, 10240, 1280))):
        run("gemm", nm, dims, GG, geglu=True)


if "--import-only" not in sys.argv:
    main()
