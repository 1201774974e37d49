"""Which resource bounds the GEMM main loop?  Times a few step shapes with the measurement-only library variants built by
scripts/probe_gemm.sh (global->LDS refills / LDS fragment reads / MFMAs removed one at a time; results are garbage).

    python scripts/probe_gemm.py            (spawns one child per variant: the library is chosen at import time)
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
VARIANTS = [(0, "full"), (1, "no-refill"), (2, "no-ldsread"), (4, "no-mfma"), (3, "mfma-only"), (5, "ldsread-only"),
            (6, "refill-only"), (7, "barriers-only")]
SHAPES = [  # (M, N, K, geglu, tile)
    (8192, 8192, 8192, False, 5),
    (4096, 4096, 4096, False, 5),
    (24576, 2560, 320, True, 5),
    (6144, 5120, 640, True, 5),
    (24576, 320, 1280, False, 12),
    (24576, 320, 320, False, 7),
    (6144, 640, 640, False, 7),
    (1536, 1280, 1280, False, 8),
]


def child():
    import torch
    sys.path.insert(0, str(ROOT))
    from seervideoldm_amd import ops
    dev = torch.device("cuda:0")
    bf16 = torch.bfloat16
    out = []
    for (M, N, K, geglu, tile) in SHAPES:
        a = torch.randn(M, K, device=dev).to(bf16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf16)
        o = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=bf16)
        fn = lambda: ops.gemm(a, w, geglu=geglu, out=o, tile=tile, splits=1)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / n * 1e3)
    print(" ".join(f"{t:9.1f}" for t in out))


def main():
    print("variant".ljust(16) + " ".join(f"{M}x{N}x{K}{'g' if g else ''}".rjust(9)[-9:] for (M, N, K, g, _) in SHAPES) + "   (us)")
    for bits, name in VARIANTS:
        env = dict(os.environ)
        if bits:
            env["SEER_HIP_LIB"] = str(ROOT / "seervideoldm_amd" / "lib" / f"libseer_probe{bits}.so")
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.strip() and l.strip()[0].isdigit()]
        print(name.ljust(16) + (line[-1] if line else "FAILED " + r.stderr[-300:]))


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else main()
