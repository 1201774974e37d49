"""Stand-alone driver for rocprofv3 --pmc passes: a handful of launches of the dominant kernel at step shapes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 scripts/pmc_gemm.py [case]
case: plain (64x64 register-staged GEMM), ring (LDS-direct 128x128), conv (128x160 conv 320->320 @32^2), all
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = torch.device("cuda:0")
case = sys.argv[1] if len(sys.argv) > 1 else "all"

if case in ("plain", "all"):
    a = torch.randn(24576, 320, device=dev).to(bf16)
    w = torch.randn(320, 320, device=dev).to(bf16)
    for _ in range(3):
        ops.gemm(a, w, tile=2, splits=1)
if case in ("ring", "all"):
    a = torch.randn(6144, 640, device=dev).to(bf16)
    w = torch.randn(5120, 640, device=dev).to(bf16)
    for _ in range(3):
        ops.gemm(a, w, tile=5, splits=1)
if case in ("conv", "all"):
    x = torch.randn(24 * 32 * 32, 320, device=dev).to(bf16)
    w = torch.randn(320, 9 * 320, device=dev).to(bf16)
    for _ in range(3):
        ops.conv3x3(x, w, 24, 32, 32)
torch.cuda.synchronize()
print("pmc driver done", case)
