"""How far does a relative input perturbation of 1e-7 .. 1e-4 move the output of one full-size denoising step?  (The bf16
pipeline rounds after every layer: any difference in fp32 summation order decorrelates the roundings of ~300 dependent layers.)
Also: the step with GroupNorm statistics from column sums against the two-stage statistics.

    python scripts/exp_perturbation.py > gpurun_out/perturbation.log
"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import SeerUNet, synth
dev = torch.device('cuda:0')
cfg = dict(synth.SD15_UNET_CFG)
m = SeerUNet(**cfg)
sd = synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev)
m = m.to(dev); m.load_state_dict(sd, strict=True); del sd
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 12, 32, 32), generator=g).to(dev)
c = torch.randn((2, 12, 77, 768), generator=g).to(dev)
t = torch.tensor([981, 981], device=dev)
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
y0 = m(x, t, c, cond_frame=2)
y0b = m(x, t, c, cond_frame=2)
print('rerun identical:', rel(y0b, y0))
for eps in (1e-7, 1e-6, 1e-5, 1e-4):
    xp = x * (1 + eps * torch.randn_like(x))
    print('input perturbed by', eps, '->', rel(m(xp, t, c, cond_frame=2), y0))
eng = m._engine
eng.gn_colsums = False
y1 = m(x, t, c, cond_frame=2)
print('two-stage vs colsums:', rel(y1, y0))
for eps in (1e-7, 1e-5):
    xp = x * (1 + eps * torch.randn_like(x))
    print('two-stage, input perturbed by', eps, '->', rel(m(xp, t, c, cond_frame=2), y1))
