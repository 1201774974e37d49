"""dW product: seer_gemm_tn_f32 vs (2 x seer_transpose_bf16 + the forward GEMM) on the training step's shapes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops, train_ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(12288, 960, 320), (12288, 320, 320), (10240, 2560, 320), (10240, 320, 1280), (3072, 1920, 640), (3072, 640, 640),
          (2560, 5120, 640), (2560, 640, 2560), (768, 3840, 1280), (768, 1280, 1280), (640, 10240, 1280), (640, 1280, 5120),
          (192, 1280, 1280), (924, 2304, 768), (924, 768, 768), (924, 6144, 768), (924, 768, 3072), (98304, 960, 320)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, K in shapes:
    dy = torch.randn((M, N), device=dev).to(torch.bfloat16)
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    out = torch.empty((N, K), device=dev)
    t_tn = timeit(lambda: train_ops.gemm_tn(dy, x, out=out))
    t_nt = timeit(lambda: ops.gemm(train_ops.transpose(dy), train_ops.transpose(x), out=out))
    fl = 2.0 * M * N * K
    print(f"M{M:6d} N{N:5d} K{K:5d}  tn {t_tn:7.1f} us ({fl / t_tn / 1e6:6.1f} TF/s)   transpose+nt {t_nt:7.1f} us ({fl / t_nt / 1e6:6.1f} TF/s)")
