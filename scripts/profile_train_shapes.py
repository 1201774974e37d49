"""Per-shape GEMM / conv / attention-forward time inside one eager fine-tuning step (config 5), HIP events per launch."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from scripts.bench_train import build  # noqa: E402
from seervideoldm_amd.profiler import TimedOps  # noqa: E402
from seervideoldm_amd.trainer import SeerTrainer  # noqa: E402

dev = torch.device("cuda:0")
unet, fst = build(dev)
fst.set_numframe(12)
top = TimedOps()
tr = SeerTrainer(unet, fst, lr=1e-5, max_grad_norm=0.3, ops=top)
g = torch.Generator().manual_seed(0)
x = torch.randn((1, 4, 12, 32, 32), generator=g).to(dev)
noise = torch.randn((1, 4, 10, 32, 32), generator=g).to(dev)
text = torch.randn((1, 77, 768), generator=g).to(dev)
t = torch.tensor([500], device=dev)
for i in range(3):
    if i == 2:
        top.reset()
    tr.forward_backward(x, noise, t, text, 2)
    torch.cuda.synchronize()
print(top.summary())
rows = top.shape_summary()
tot = sum(r[2] for r in rows)
print(f"total timed {tot:.2f} ms over {sum(r[1] for r in rows)} launches")
for tag, n, ms, tf in rows[:60]:
    print(f"{ms:8.3f} ms  {n:4d}x  {tf:7.1f} TF/s  {tag}")
