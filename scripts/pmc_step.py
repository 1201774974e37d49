"""Stand-alone driver for rocprofv3 --pmc passes over ONE full-size eager denoising step (no hipGraph, no events).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 scripts/pmc_step.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from seervideoldm_amd import DDIMSampler, SeerUNet, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(synth.SD15_UNET_CFG)
model = SeerUNet(**cfg).to(dev)
model.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev), strict=True)
x_T, x0_emb, c, uc = bench.build_inputs(dev)
smp = DDIMSampler(dev)
smp.make_schedule(50, verbose=False)
ts = smp._t_table[49].expand(1)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for _ in range(nsteps):
    smp.p_sample_ddim(model, x_T, c, ts, index=49, x0_emb=x0_emb, unconditional_guidance_scale=7.5,
                      unconditional_conditioning=uc)
torch.cuda.synchronize()
print("pmc step driver done")
