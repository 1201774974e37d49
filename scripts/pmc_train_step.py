"""Stand-alone driver for rocprofv3 --pmc passes over eager full-size fine-tuning steps (config 5; no hipGraph, no events).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 scripts/pmc_train_step.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from scripts.bench_train import build  # noqa: E402
from seervideoldm_amd.trainer import SeerTrainer  # noqa: E402

dev = torch.device("cuda:0")
unet, fst = build(dev)
fst.set_numframe(12)
tr = SeerTrainer(unet, fst, lr=1e-5, max_grad_norm=0.3)
g = torch.Generator().manual_seed(0)
x = torch.randn((1, 4, 12, 32, 32), generator=g).to(dev)
noise = torch.randn((1, 4, 10, 32, 32), generator=g).to(dev)
text = torch.randn((1, 77, 768), generator=g).to(dev)
t = torch.tensor([500], device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    tr.forward_backward(x, noise, t, text, 2)
    tr.optimizer_step()
torch.cuda.synchronize()
print("pmc train step driver done")
