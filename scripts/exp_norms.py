"""LayerNorm and GroupNorm-apply launches on the engine's shapes inside a replayed HIP graph (20 launches per replay): us per
launch and the bytes moved per second.  SEER_HIP_LIB=<other build> runs the same against another library build.

    python scripts/exp_norms.py > gpurun_out/norms.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402
from exp_gn_colsums import graph_us  # noqa: E402

bf16 = torch.bfloat16


def main():
    dev = torch.device("cuda:0")
    print("layernorm rows x C            us      TB/s")
    for rows, C in [(24576, 320), (20480, 320), (6144, 640), (5120, 640), (1536, 1280), (384, 1280)]:
        x = torch.randn((rows, C), device=dev).to(bf16)
        g, b = torch.ones((C,), device=dev), torch.zeros((C,), device=dev)
        t = graph_us(lambda: ops.layernorm(x, g, b))
        print(f"{rows:6d} x {C:5d}         {t:8.2f}  {4 * rows * C / t * 1e-6:8.2f}")
    print("groupnorm apply B x rows x (C1+C2)     us      TB/s")
    B, G = 2, 32
    for rows, C1, C2 in [(12288, 320, 0), (12288, 640, 0), (12288, 640, 320), (3072, 640, 0), (3072, 1280, 640),
                         (768, 1280, 0), (768, 1280, 1280), (192, 1280, 0), (192, 1280, 1280)]:
        x1 = torch.randn((B * rows, C1), device=dev).to(bf16)
        x2 = torch.randn((B * rows, C2), device=dev).to(bf16) if C2 else None
        Ct = C1 + C2
        stats = torch.zeros((B, G, 2), device=dev)
        ops.groupnorm_stats(x1, x2, B, G, stats)
        gm, bt = torch.ones((Ct,), device=dev), torch.zeros((Ct,), device=dev)
        t = graph_us(lambda: ops.groupnorm_apply(x1, x2, B, G, stats, rows * (Ct // G), 1e-5, gm, bt, True))
        print(f"{B} x {rows:6d} x ({C1}+{C2})      {t:8.2f}  {4 * B * rows * Ct / t * 1e-6:8.2f}")


if __name__ == "__main__":
    main()
