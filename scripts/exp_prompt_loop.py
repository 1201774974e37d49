"""eval-style loop at full size: a NEW prompt for every clip (50 DDIM steps, no decode), hipGraph replay.  Per-clip wall time."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import DDIMSampler, SeerUNet, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(synth.SD15_UNET_CFG)
m = SeerUNet(**cfg).to(dev)
m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev), strict=True)
m.eval()
m.use_graph = True
smp = DDIMSampler(dev)
g = torch.Generator().manual_seed(0)
x0 = (torch.randn((1, 4, 2, 32, 32), generator=g) * 0.9).to(dev)
uc = torch.randn((1, 1, 77, 768), generator=g).expand(-1, 12, -1, -1).contiguous().to(dev)
times = []
for i in range(6):
    c = torch.randn((1, 12, 77, 768), generator=g).to(dev)
    xT = torch.randn((1, 4, 10, 32, 32), generator=g).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat, _ = smp.sample(unet=m, S=50, conditioning=c, batch_size=1, shape=(4, 10, 32, 32), x0_emb=x0, verbose=False,
                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=xT, is_3d=True)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    assert torch.isfinite(lat).all()
print("per-clip ms (50 steps, new prompt each):", [round(t, 1) for t in times])
