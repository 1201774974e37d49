"""TEST INFRASTRUCTURE ONLY -- a checkpoint directory written by the REAL `accelerate.Accelerator.save_state` (accelerate is installed
in the build container) around the REAL reference modules, as train.py:395-399 writes it, plus the sidecar of train.py:399.

    python -m oracle.make_goldens_accel            (build container only; ~20 s)

What runs: the reference's `SeerUNet` + `FSTextTransformer` (oracle/ref_import.py) at a 32-channel width in train mode with the
trainable set of train.py:188-192 (temporal_attentions + FSTextTransformer), `torch.optim.AdamW` over train.py:213's parameter list,
a LambdaLR cosine-with-warmup schedule (what diffusers' `get_scheduler("cosine")` builds), `accelerator.prepare(...)` of all four,
TWO real optimizer steps of the loss of train.py:343-387 (so the moments, the step counters and the scheduler have moved), then
`accelerator.save_state(dir, safe_serialization=False)` -- the file names of the accelerate the reference pins (pytorch_model.bin,
pytorch_model_1.bin, optimizer.bin, scheduler.bin, random_states_0.pkl) -- and `accelerator.save({...}, dir + ".pt")`.

Output: tests/golden/accelerate_state/learned_sdunet-steps-2/ + learned_sdunet-steps-2.pt + expected.npz (what a loader must end up
with: per-name fingerprints of the weights and of both Adam moments in the REFERENCE's parameter order, the step count, the meters).
Every file is data a training run writes; none of the reference's source is stored.  tests/test_checkpoint.py loads the directory
with seervideoldm_amd.checkpoint.load_checkpoint.
"""
from __future__ import annotations

import math
import shutil
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import ref_import  # noqa: E402
from seervideoldm_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden" / "accelerate_state"
CFG = dict(block_out_channels=(32, 32, 32, 32), layers_per_block=1, cross_attention_dim=40, attention_head_dim=8)
FS = dict(num_frames=4, num_layers=1, channels=40, n_heads=1, cross_attention_dim=40)       # head dim 40: one the product instantiates
HP = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
WARMUP, TOTAL = 1, 10
STEPS = 2


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def batch(step):
    """seeded inputs of optimizer step `step` (tests re-draw them)"""
    x = _randn((1, 4, 3, 8, 8), 100 + step)              # 1 conditioning + 2 predicted frames, 8x8 latent
    noise = _randn((1, 4, 2, 8, 8), 200 + step)
    text = _randn((1, 77, FS["channels"]), 300 + step)
    return x, noise, text, torch.tensor([417 - 100 * step])


def fingerprint(t: torch.Tensor) -> np.ndarray:
    """(sum, sum of squares, first, last) in float64: enough to tell any two tensors of a checkpoint apart"""
    f = t.detach().double().flatten()
    return np.asarray([f.sum().item(), (f * f).sum().item(), f[0].item(), f[-1].item()])


def main():
    from accelerate import Accelerator
    ref = ref_import.load_reference()
    torch.manual_seed(0)
    unet = ref.unet.SeerUNet(**CFG)
    ref_import.enable_xformers_path(unet)
    unet.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(CFG)), strict=True)
    fst = ref.unet.FSTextTransformer(num_frames=FS["num_frames"], in_channels=FS["channels"], out_channels=FS["channels"],
                                     n_heads=FS["n_heads"], num_layers=FS["num_layers"], cross_attention_dim=FS["cross_attention_dim"])
    ref_import.enable_xformers_path(fst)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**FS)), strict=True)
    fst.set_numframe(3)
    unet.requires_grad_(False)                                            # train.py:188-192
    for name, module in unet.named_modules():
        if name.endswith(("temporal_attentions",)):
            for prm in module.parameters():
                prm.requires_grad = True
    unet.train(); fst.train()
    params = list(filter(lambda p: p.requires_grad, unet.parameters())) + list(fst.parameters())      # train.py:213
    names = [k for k, p in unet.named_parameters() if p.requires_grad] + ["fstext:" + k for k, _ in fst.named_parameters()]
    opt = torch.optim.AdamW(params, **HP)

    def lr_lambda(step):            # diffusers.optimization.get_cosine_schedule_with_warmup
        if step < WARMUP:
            return float(step) / float(max(1, WARMUP))
        progress = float(step - WARMUP) / float(max(1, TOTAL - WARMUP))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)
    acc = Accelerator(cpu=True)
    unet, fst, opt, sched = acc.prepare(unet, fst, opt, sched)

    class Meter:                    # the sidecar's two dicts are {"vals", "avg", "steps"} (train.py:76-77)
        def __init__(self):
            self.vals, self.steps, self.val, self.avg = [], [], None, 0

        def update(self, val, step):
            self.avg = val if self.val is None else self.avg * 0.99 + val * 0.01
            self.val = val
            self.vals.append(val)
            self.steps.append(step)

        def ckpt(self):
            return {"vals": self.vals, "avg": self.avg, "steps": self.steps}
    lr_meter, losses = Meter(), Meter()
    global_step = 0
    for s in range(STEPS):
        x, noise, text, t = batch(s)
        text_seq = fst(context=text)
        pred = unet(x, t, text_seq, 1)
        loss = torch.nn.functional.mse_loss(pred[:, :, 1:], noise, reduction="none").mean([1, 2, 3, 4]).mean()
        acc.backward(loss)
        acc.clip_grad_norm_(params, 0.3)
        opt.step(); sched.step(); opt.zero_grad()
        losses.update(float(loss), global_step)
        lr_meter.update(sched.get_last_lr()[0], global_step)
        global_step += 1
    if OUT.exists():
        shutil.rmtree(OUT)
    OUT.mkdir(parents=True)
    save_path = OUT / f"learned_sdunet-steps-{global_step}"
    acc.save_state(str(save_path), safe_serialization=False)                                            # train.py:396-397
    acc.save({"epoch": 0, "global_step": global_step, "lr_meter": lr_meter.ckpt(), "losses_train": losses.ckpt()},
             str(save_path) + ".pt")                                                                    # train.py:398-399
    raw = acc.unwrap_model(unet), acc.unwrap_model(fst)
    osd = opt.state_dict()
    exp = dict(param_names=np.asarray(names), global_step=np.int64(global_step), last_lr=np.float64(sched.get_last_lr()[0]),
               unet_keys=np.asarray(list(raw[0].state_dict())), fstext_keys=np.asarray(list(raw[1].state_dict())),
               unet_fp=np.stack([fingerprint(v) for v in raw[0].state_dict().values()]),
               fstext_fp=np.stack([fingerprint(v) for v in raw[1].state_dict().values()]),
               exp_avg_fp=np.stack([fingerprint(osd["state"][i]["exp_avg"]) for i in range(len(names))]),
               exp_avg_sq_fp=np.stack([fingerprint(osd["state"][i]["exp_avg_sq"]) for i in range(len(names))]),
               opt_step=np.float64(float(osd["state"][0]["step"])), loss_vals=np.asarray(losses.vals), lr_vals=np.asarray(lr_meter.vals))
    np.savez_compressed(OUT / "expected.npz", **exp)
    for p in sorted(OUT.rglob("*")):
        if p.is_file():
            print(f"{p.relative_to(OUT)}  {p.stat().st_size} bytes")
    print(f"accelerate {__import__('accelerate').__version__}, torch {torch.__version__}; losses {losses.vals}")


if __name__ == "__main__":
    main()
