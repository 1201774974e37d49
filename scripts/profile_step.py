"""Per-shape time table of one full-size denoising step (HIP-event timed launches). Run on the GPU box.

    python scripts/profile_step.py [guidance_scale=7.5] [frames=12]
scale 1.0 drops the unconditional half (B = 1: the per-rank work of a 2-GPU CFG split); fewer frames approximate a frame shard."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from seervideoldm_amd import DDIMSampler, SeerUNet, synth  # noqa: E402
from seervideoldm_amd.profiler import TimedOps  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(synth.SD15_UNET_CFG)
model = SeerUNet(**cfg).to(dev)
model.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev), strict=True)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 7.5
frames = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOAD["frames"]
bench.WORKLOAD["frames"] = frames
x_T, x0_emb, c, uc = bench.build_inputs(dev)
smp = DDIMSampler(dev)
smp.make_schedule(50, verbose=False)
ts = smp._t_table[49].expand(1)
step = lambda: smp.p_sample_ddim(model, x_T, c, ts, index=49, x0_emb=x0_emb, unconditional_guidance_scale=scale,
                                 unconditional_conditioning=uc)
step()
timed = TimedOps()
model._engine.ops = timed
step()
torch.cuda.synchronize()
timed.reset()
bench.gpu_busy(60.0, dev)
reps = 3
for _ in range(reps):
    step()
torch.cuda.synchronize()
tot = 0.0
print(f"{'shape':58s} {'calls':>5s} {'ms/step':>8s} {'us/call':>8s} {'TF/s':>7s}")
for tag, n, ms, tf in timed.shape_summary():
    tot += ms / reps
    print(f"{tag:58s} {n // reps:5d} {ms / reps:8.3f} {ms / n * 1e3:8.1f} {tf:7.1f}")
print(f"total {tot:.3f} ms/step")
for k, v in timed.summary().items():
    print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()})
