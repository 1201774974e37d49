"""The oracle's training step (oracle/seer_oracle.py: train_loss_and_grads, clip_and_adamw) against the REFERENCE's own
step: tests/golden/train_tiny.npz was produced by oracle/make_goldens.py::gen_train from the real SeerUNet / FSTextTransformer
modules, torch autograd, torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW (train.py:319-389)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import seer_oracle as O
from seervideoldm_amd import synth

G = Path(__file__).parent / "golden" / "train_tiny.npz"
UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(32, 64, 64, 64), cross_attention_dim=64,
            attention_head_dim=8, layers_per_block=2)
FSTEXT = dict(num_frames=16, num_layers=1, channels=64, n_heads=2, cross_attention_dim=64)
HP = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=0.3)


@pytest.fixture(scope="module")
def step():
    g = np.load(G, allow_pickle=False)
    T = lambda k: torch.from_numpy(g[k])
    usd = synth.synth_state_dict(synth.unet_param_shapes(UNET))
    fsd = synth.synth_state_dict(synth.fstext_param_shapes(**FSTEXT))
    loss, gu, gf, pred = O.train_loss_and_grads(usd, {**O.DEFAULT_CFG, **UNET}, fsd, T("model_input"), T("noise"), T("timestep"),
                                                T("text"), 2, fstext_heads=FSTEXT["n_heads"])
    return g, usd, fsd, loss, gu, gf, pred


def test_add_noise_matches_scheduler(step):
    g = step[0]
    T = lambda k: torch.from_numpy(g[k])
    a = T("alphas_cumprod")[T("timestep")].reshape(-1, 1, 1, 1, 1)
    x = torch.cat([T("latents_x0"), a.sqrt() * T("latents") + (1 - a).sqrt() * T("noise")], 2)
    assert torch.allclose(x, T("model_input"), atol=1e-6)


def test_loss_and_prediction(step):
    g, _, _, loss, _, _, pred = step
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    assert np.abs(pred.numpy() - g["pred"]).max() < 2e-4


def test_every_gradient_statistic(step):
    g, _, _, _, gu, gf, _ = step
    for keys, stats, grads in ((g["unet_keys"], g["unet_grad_stats"], gu), (g["fstext_keys"], g["fstext_grad_stats"], gf)):
        assert set(map(str, keys)) == set(grads), "trainable parameter set differs from the reference's"
        for k, (norm, total) in zip(map(str, keys), stats):
            gr = grads[k]
            assert abs(float(gr.norm()) - norm) <= 2e-4 * max(norm, 1e-3) + 1e-7, (k, float(gr.norm()), norm)
            assert abs(float(gr.sum()) - total) <= 2e-3 * max(norm, 1e-3) * gr.numel() ** 0.5 + 1e-6, (k, float(gr.sum()), total)


def test_full_gradients_and_adamw_update(step):
    g, usd, fsd, _, gu, gf, _ = step
    for pre, grads in (("gu:", gu), ("gf:", gf)):
        for name in g.files:
            if name.startswith(pre):
                ref = torch.from_numpy(g[name])
                got = grads[name[3:]]
                assert (got - ref).norm() <= 1e-3 * ref.norm() + 1e-7, name
    # clip (UNet parameters only, train.py:384) + AdamW step 1
    pu = {k: usd[k].clone().float() for k in gu}
    pf = {k: fsd[k].clone().float() for k in gf}
    z = lambda d: {k: torch.zeros_like(v) for k, v in d.items()}
    total = O.clip_and_adamw(pu, gu, z(pu), z(pu), 1, HP["lr"], HP["betas"], HP["eps"], HP["weight_decay"], HP["max_grad_norm"])
    O.clip_and_adamw(pf, gf, z(pf), z(pf), 1, HP["lr"], HP["betas"], HP["eps"], HP["weight_decay"], None)
    assert abs(float(total) - float(g["unet_grad_norm"])) < 1e-4 * float(g["unet_grad_norm"])
    for pre, params in (("pu:", pu), ("pf:", pf)):
        for name in g.files:
            if name.startswith(pre):
                ref = torch.from_numpy(g[name])
                assert (params[name[3:]] - ref).abs().max() < 2e-5, name


def test_text_loss_variant():
    """`text_loss: True` (train.py:346-347,377-378): loss_text and the FSTextTransformer gradients of the reference's step"""
    g = np.load(G, allow_pickle=False)
    T = lambda k: torch.from_numpy(g[k])
    usd = synth.synth_state_dict(synth.unet_param_shapes(UNET))
    fsd = synth.synth_state_dict(synth.fstext_param_shapes(**FSTEXT))
    loss, _, gf, _ = O.train_loss_and_grads(usd, {**O.DEFAULT_CFG, **UNET}, fsd, T("model_input"), T("noise"), T("timestep"),
                                            T("text"), 2, fstext_heads=FSTEXT["n_heads"], text_loss=True)
    ref = float(g["loss"]) + float(g["loss_text"])
    assert abs(float(loss) - ref) < 1e-5 * max(1.0, ref)
    for k, (norm, total) in zip(map(str, g["fstext_keys"]), g["fstext_grad_stats_text_loss"]):
        assert abs(float(gf[k].norm()) - norm) <= 2e-4 * max(norm, 1e-3) + 1e-7, (k, float(gf[k].norm()), norm)
    k = "trf_blocks.0.transformer_blocks.1.attn1.to_q.weight"
    r = torch.from_numpy(g["gft:" + k])
    assert (gf[k] - r).norm() <= 1e-3 * r.norm()
