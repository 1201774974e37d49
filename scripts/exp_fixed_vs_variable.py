"""Fixed against variable cost of every kernel family of a denoising step (review item 1 of round 5): the same network evaluated at
the step's shape (CFG batch 2 x 12 frames x 32x32 latent) and at a shape whose launches hold almost no rows (B = 1, one frame, 8x8
latent: 64 / 16 / 4 / 1 rows at the four levels), every launch bracketed by HIP events on the launch stream
(seervideoldm_amd/profiler.py).  A family's time at the small shape is what its launches cost whatever they compute -- dispatch,
prologue, the K loop over the (unchanged) weights, epilogue -- its "fixed" part; the rest scales with the rows.

    python scripts/exp_fixed_vs_variable.py > profiles/r06_fixed_vs_variable.md
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import SeerUNet, synth  # noqa: E402
from seervideoldm_amd import ops as plain_ops  # noqa: E402
from seervideoldm_amd.profiler import TimedOps  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(synth.SD15_UNET_CFG)
m = SeerUNet(**cfg).to(dev)
m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev), strict=True)
m.eval()
REPS = 3


def busy(ms):
    a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    out = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(max(1, int(ms / 1.3))):
        plain_ops.gemm(a, a, out=out, tile=1)


def families(B, Fr, h):
    """family -> [launches per evaluation, us per launch, ms per evaluation]; families are named by LEVEL (rows of the launch mapped
    through this shape's own level table), so the two shapes line up"""
    x = torch.randn((B, 4, Fr, h, h), device=dev)
    c = torch.randn((B, Fr, 77, 768), device=dev)
    t = torch.tensor([981] * B, device=dev)
    m.prepare()
    eng = m._engine
    timed = TimedOps()
    timed.level_of_rows = {B * Fr * h * h: "L0", B * Fr * h * h // 4: "L1", B * Fr * h * h // 16: "L2", B * Fr * h * h // 64: "L3/mid"}
    eng.ops = timed
    m(x, t, c, cond_frame=0)
    torch.cuda.synchronize()
    timed.reset()
    busy(60.0)
    for _ in range(REPS):
        m(x, t, c, cond_frame=0)
    torch.cuda.synchronize()
    out = {}
    for cls, recs in timed.records.items():
        for r in recs:
            name = timed._family(cls, r[4], timed.level_of_rows)
            if name.startswith("conv3x3 ") and name.split()[1].split("x")[0].isdigit():     # "conv3x3 32x32" -> by level
                side = int(name.split()[1].split("x")[0])
                name = "conv3x3 " + {h: "L0", h // 2: "L1", h // 4: "L2", h // 8: "L3/mid"}.get(side, name.split()[1])
            if name.startswith("attention"):                # "... d40 Sq1024" -> by level (query rows of the launch)
                tg = r[4].split()
                name = " ".join(name.split()[:-1]) + " " + timed.level_of_rows.get(int(tg[1][1:]) * int(tg[2][2:]), "?")
            a = out.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += r[0].elapsed_time(r[1])
    return {k: (n // REPS, ms / n * 1e3, ms / REPS) for k, (n, ms) in out.items()}


full = families(2, 12, 32)
tiny = families(1, 1, 8)
print("# Fixed against variable cost per kernel family (HIP events on the launch stream, eager launches, 3 evaluations each)")
print()
print("`step shape` = CFG batch 2 x 12 frames x 32x32 latent (24 576 / 6 144 / 1 536 / 384 rows at the four levels); `no rows` = B 1, one")
print("frame, 8x8 latent (64 / 16 / 4 / 1 rows): the same launches over the same weights with next to nothing to compute.  The fused")
print("feed-forward is not taken at the small shape (its three launches are), so the L0 feed-forward rows compare different kernels.")
print()
print("| family | launches | us / launch, step shape | us / launch, no rows | ms / step | of which fixed, ms | fixed share |")
print("|---|---:|---:|---:|---:|---:|---:|")
tot = [0.0, 0.0, 0]
for name, (n, us, ms) in sorted(full.items(), key=lambda kv: -kv[1][2]):
    tn = tiny.get(name)
    if tn is None:
        print(f"| {name} | {n} | {us:.1f} | - | {ms:.3f} | - | - |")
        tot[0] += ms
        tot[2] += n
        continue
    fixed = min(tn[1], us) * n * 1e-3
    print(f"| {name} | {n} | {us:.1f} | {tn[1]:.1f} | {ms:.3f} | {fixed:.3f} | {fixed / ms:.0%} |")
    tot[0] += ms
    tot[1] += fixed
    tot[2] += n
print(f"| **total** | {tot[2]} | | | {tot[0]:.3f} | {tot[1]:.3f} | {tot[1] / tot[0]:.0%} |")
print()
only_tiny = sorted(set(tiny) - set(full))
if only_tiny:
    print("families of the small shape only: " + "; ".join(f"{k}: {tiny[k][0]} x {tiny[k][1]:.1f} us" for k in only_tiny))
