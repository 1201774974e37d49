"""What ONE rank of a partitioned step computes, on one GPU: the full-size UNet at the (CFG batch, frames) a rank of the
north star's partition holds -- batch groups first (B = 1 per rank from 2 GPUs on), then frame shards (12 frames / 2, 4, 6).
Same kernels and schedule as the sharded engine minus the exchanges (GroupNorm statistics from column sums in both).

    python scripts/exp_shard_sizes.py > gpurun_out/shard_sizes.log
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import SeerUNet, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(synth.SD15_UNET_CFG)
m = SeerUNet(**cfg)
sd = synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev)
m = m.to(dev)
m.load_state_dict(sd, strict=True)
del sd
m.use_graph = True
def evaluate(B, Fr):
    x = torch.randn((B, 4, Fr, 32, 32), device=dev)
    c = torch.randn((B, Fr, 77, 768), device=dev)
    t = torch.tensor([981] * B, device=dev)
    for _ in range(3):
        m(x, t, c, cond_frame=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m(x, t, c, cond_frame=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 50


print("config 2 (Sthv2, CFG batch 2 x 12 frames): batch x CFG groups first, then frame shards")
print("ranks  partition                 per-rank shape       ms per UNet evaluation   speed-up over one GPU (compute only)")
base = None
for ranks, part, B, Fr in [(1, "1 x 1", 2, 12), (2, "2 batch groups", 1, 12), (4, "2 groups x 2 frame shards", 1, 6),
                           (8, "2 groups x 4 frame shards", 1, 3), (12, "2 groups x 6 frame shards", 1, 2)]:
    ms = evaluate(B, Fr)
    base = base or ms
    print(f"{ranks:5d}  {part:26s} B={B} F={Fr:2d} 32x32      {ms:8.2f}                 {base / ms:5.2f}x")
print("config 3 (Bridge, CFG batch 8 x 16 frames): 8 batch groups, no frame shards, no per-layer exchange")
base = None
for ranks, part, B, Fr in [(1, "1 x 1", 8, 16), (2, "2 batch groups", 4, 16), (4, "4 batch groups", 2, 16), (8, "8 batch groups", 1, 16)]:
    ms = evaluate(B, Fr)
    base = base or ms
    print(f"{ranks:5d}  {part:26s} B={B} F={Fr:2d} 32x32      {ms:8.2f}                 {base / ms:5.2f}x")
# config 5 (fine-tuning step, b = 1 per rank, data parallel): every rank runs the b = 1 step; the compute-only ceiling of N ranks is
# N x (one b = 1 step) against ONE GPU stepping through the same N samples at its best micro-batch
from scripts.bench_train import time_train  # noqa: E402

m.use_graph = False
t1 = time_train(dev, steps=5, warmup=2, unet=m)["ms_per_step"]
t8 = time_train(dev, steps=3, warmup=1, b=8, unet=m)["ms_per_step"]
print("config 5 (fine-tuning step, 12 frames, 2 conditioning): data parallel, one gradient all-reduce per step")
print(f"    b = 1 step {t1:.2f} ms = {1e3 / t1:.1f} samples/s per rank; one GPU at micro-batch 8: {t8:.2f} ms = {8e3 / t8:.1f} samples/s")
print(f"    8 ranks at b = 1 (compute only): {8e3 / t1:.1f} samples/s = {t8 / t1:.2f}x one GPU at micro-batch 8, "
      f"{8.0:.1f}x one GPU at b = 1")
