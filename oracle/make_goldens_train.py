"""TEST INFRASTRUCTURE ONLY -- fixtures for the fine-tuning step of tests/test_gpu_train.py (train.py:343-387), so that the GPU box's
host cores no longer run fp32 autograd (43 s for the wide case alone, 2.3 x that on the driver's box).

    python -m oracle.make_goldens_train            (build container only; ~4 min)

Per case: the loss, the prediction (fp32, whole) and, per trainable tensor, a SKETCH of its gradient: the squared norm and K = 4 inner
products with seeded standard-normal vectors (oracle/sketch.py: torch.Generator seeded by the CRC of the tensor's name).  For an
error e = g_hip - g_ref, E[(e . r)^2] = |e|^2, so mean_j (g_hip . r_j - c_j)^2 estimates |e|^2 per tensor (+-70 %) and over the 227
tensors of a case (+-5 %): enough for the relative-L2 bounds the test has always asserted (4e-2 overall), in 15 KB instead of the
90-600 MB of the gradients themselves.  Every fixture is data.

Source of the numbers: the constant-width cases run the REAL reference (oracle/ref_import.py: SeerUNet + FSTextTransformer in
train mode, temporal_attentions + FSTextTransformer trainable, torch autograd).  The case at the real widths with ONE layer per
block is not a configuration the reference can build (unet_3d_blocks.py:311,354 takes the downsampler width from the loop
variable); it runs oracle/seer_oracle.py::train_loss_and_grads, the restatement that tests/test_train_oracle.py pins to the
reference's step, and tests/test_oracle_golden.py checks the restatement against the reference-made cases of THIS file.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import ref_import, seer_oracle as O  # noqa: E402
from oracle.make_goldens import _save  # noqa: E402
from oracle.sketch import K_SKETCH, sketch  # noqa: E402,F401
from seervideoldm_amd import synth  # noqa: E402

CFG_MINI = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
CFG_WIDE = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
FS = dict(num_frames=16, num_layers=1, channels=192, n_heads=2, cross_attention_dim=192)
# (name, UNet config, B, frames, conditioning frames, latent side): the parameters of test_train_step_matches_oracle
CASES = [("mini_1_3_1_16", CFG_MINI, 1, 3, 1, 16), ("mini_2_4_2_8", CFG_MINI, 2, 4, 2, 8), ("mini_1_3_1_32", CFG_MINI, 1, 3, 1, 32),
         ("wide_1_4_2_16", CFG_WIDE, 1, 4, 2, 16)]


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def inputs(B, Fr, cond, H):
    """the seeded inputs of the test (tests/test_gpu_train.py)"""
    x, noise = _randn((B, 4, Fr, H, H), 1), _randn((B, 4, Fr - cond, H, H), 2)
    text, t = _randn((B, 77, 192), 3), torch.tensor([417, 93, 800, 5][:B])
    return x, noise, text, t


def reference_step(ref, cfg, B, Fr, cond, H):
    unet = ref.unet.SeerUNet(**cfg)
    ref_import.enable_xformers_path(unet)
    unet.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg)), strict=True)
    fst = ref.unet.FSTextTransformer(num_frames=FS["num_frames"], in_channels=FS["channels"], out_channels=FS["channels"],
                                     n_heads=FS["n_heads"], num_layers=FS["num_layers"], cross_attention_dim=FS["cross_attention_dim"])
    ref_import.enable_xformers_path(fst)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**FS)), strict=True)
    fst.set_numframe(Fr)                                                  # train.py:187
    unet.requires_grad_(False)                                            # train.py:188-192
    for name, module in unet.named_modules():
        if name.endswith(("temporal_attentions",)):
            for prm in module.parameters():
                prm.requires_grad = True
    unet.train(); fst.train()
    x, noise, text, t = inputs(B, Fr, cond, H)
    text_seq = fst(context=text)
    pred = unet(x, t, text_seq, cond)
    loss = torch.nn.functional.mse_loss(pred[:, :, cond:], noise, reduction="none").mean([1, 2, 3, 4]).mean()     # train.py:367-380
    loss.backward()
    zero = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
    gu = {k: zero(p) for k, p in unet.named_parameters() if p.requires_grad}
    gf = {k: zero(p) for k, p in fst.named_parameters()}
    return loss.detach(), gu, gf, pred.detach()


def oracle_step(cfg, B, Fr, cond, H):
    usd = synth.synth_state_dict(synth.unet_param_shapes(cfg))
    fsd = synth.synth_state_dict(synth.fstext_param_shapes(**FS))
    x, noise, text, t = inputs(B, Fr, cond, H)
    return O.train_loss_and_grads(usd, {**O.DEFAULT_CFG, **cfg}, fsd, x, noise, t, text, cond, fstext_heads=FS["n_heads"])


def gen_unet_wide():
    """tests/test_gpu_unet.py::test_unet_forward_matches_oracle[wide-1-2-16-0]: the real widths with one layer per block (head dims
    40 / 80 / 160) -- not a configuration the reference builds; the restatement's output, so that the GPU box does not spend 26 s of
    host time on it"""
    cfg = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=1, cross_attention_dim=768, attention_head_dim=8)
    sd = synth.synth_state_dict(synth.unet_param_shapes(cfg))
    x, ctx, t = _randn((1, 4, 2, 16, 16), 1), _randn((1, 2, 77, 768), 2), torch.tensor([501])
    with torch.no_grad():
        y = O.unet_forward(sd, {**O.DEFAULT_CFG, **cfg}, x, t, ctx, cond_frame=0)
    _save("unet_wide_1_2_16_0.npz", out=y.float().numpy(), from_reference=np.int64(0))


def main():
    gen_unet_wide()
    ref = ref_import.load_reference()
    for name, cfg, B, Fr, cond, H in CASES:
        t0 = time.time()
        from_reference = cfg is CFG_MINI
        loss, gu, gf, pred = reference_step(ref, cfg, B, Fr, cond, H) if from_reference else oracle_step(cfg, B, Fr, cond, H)
        out = dict(loss=np.float64(float(loss)), pred=pred.float().numpy(), from_reference=np.int64(from_reference),
                   unet_keys=np.asarray(sorted(gu)), fstext_keys=np.asarray(sorted(gf)),
                   unet_sketch=np.stack([sketch("unet:" + k, gu[k]) for k in sorted(gu)]),
                   fstext_sketch=np.stack([sketch("fstext:" + k, gf[k]) for k in sorted(gf)]))
        _save(f"train_{name}.npz", **out)
        print(f"train_{name}.npz: loss {float(loss):.6f}, {len(gu)} + {len(gf)} tensors, "
              f"{'reference' if from_reference else 'oracle restatement'}, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
