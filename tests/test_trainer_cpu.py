"""Host-logic test (CPU) of seervideoldm_amd.trainer.SeerTrainer: its hand-written reverse schedule (what is kept, gradient
fan-in at residuals / skips, the stop point below the first trainable block, packed parameter layouts, clip + AdamW
bookkeeping) driven through plain-torch stand-ins for the kernel library must reproduce autograd of the oracle.
NOT a product path -- on a GPU box the same schedule runs on libseer_hip.so (tests/test_gpu_train.py)."""
import pytest
import torch

from oracle import seer_oracle as O
from seervideoldm_amd import FSTextTransformer, SeerUNet, synth
from seervideoldm_amd.trainer import SeerTrainer
from tests import torch_ops_backend as tob
from tests import torch_train_ops_backend as ttob

CFG = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
FS = dict(num_frames=16, num_layers=1, channels=192, n_heads=2, cross_attention_dim=192)
HP = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=0.3)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.fixture(scope="module")
def setup():
    usd = synth.synth_state_dict(synth.unet_param_shapes(CFG))
    fsd = synth.synth_state_dict(synth.fstext_param_shapes(**FS))
    unet = SeerUNet(**CFG)
    unet.load_state_dict(usd, strict=True)
    fst = FSTextTransformer(num_frames=FS["num_frames"], in_channels=192, out_channels=192, n_heads=2, num_layers=1,
                            cross_attention_dim=192)
    fst.load_state_dict(fsd, strict=True)
    return usd, fsd, unet, fst


@pytest.mark.parametrize("B,Fr,cond,H", [(1, 3, 1, 8), (2, 3, 2, 8), (1, 2, 0, 16)])
def test_trainer_schedule_matches_oracle_autograd(setup, B, Fr, cond, H):
    usd, fsd, unet, fst = setup
    fst.set_numframe(Fr)
    tr = SeerTrainer(unet, fst, ops=tob, tops=ttob, **HP)
    x = _randn((B, 4, Fr, H, H), 1)
    noise = _randn((B, 4, Fr - cond, H, H), 2)
    text = _randn((B, 77, 192), 3)
    t = torch.tensor([417, 93, 800, 5][:B])          # a different timestep per batch element
    loss = tr.forward_backward(x, noise, t, text, cond)
    ref_loss, gu, gf, pred = O.train_loss_and_grads(usd, {**O.DEFAULT_CFG, **CFG}, fsd, x, noise, t, text, cond, fstext_heads=2)
    assert abs(float(loss) - float(ref_loss)) < 2e-2 * float(ref_loss)
    got = tr.trainable_state_dict_of(tr.pu.g, tr.pf.g)
    for name, ref, mine in (("unet", gu, got["unet"]), ("fstext", gf, got["fstext"])):
        assert set(ref) == set(mine), (name, set(ref) ^ set(mine))
        num = sum(((mine[k].reshape(ref[k].shape) - ref[k]) ** 2).sum() for k in ref) ** 0.5
        den = sum((ref[k] ** 2).sum() for k in ref) ** 0.5
        assert num / den < 3e-2, (name, float(num / den))
        worst = max(ref, key=lambda k: float((mine[k].reshape(ref[k].shape) - ref[k]).norm() / (ref[k].norm() + 1e-3 * den)))
        w = float((mine[worst].reshape(ref[worst].shape) - ref[worst]).norm() / (ref[worst].norm() + 1e-3 * den))
        assert w < 0.08, (name, worst, w)
    # optimizer: clip over the UNet parameters only, AdamW on both segments
    pu = {k: usd[k].clone().float() for k in gu}
    pf = {k: fsd[k].clone().float() for k in gf}
    z = lambda d: {k: torch.zeros_like(v) for k, v in d.items()}
    mine_g = tr.trainable_state_dict_of(tr.pu.g, tr.pf.g)
    O.clip_and_adamw(pu, {k: mine_g["unet"][k].reshape(pu[k].shape) for k in pu}, z(pu), z(pu), 1, HP["lr"], HP["betas"],
                     HP["eps"], HP["weight_decay"], HP["max_grad_norm"])
    O.clip_and_adamw(pf, {k: mine_g["fstext"][k].reshape(pf[k].shape) for k in pf}, z(pf), z(pf), 1, HP["lr"], HP["betas"],
                     HP["eps"], HP["weight_decay"], None)
    tr.optimizer_step()
    new = tr.trainable_state_dict()
    for name, ref, mine in (("unet", pu, new["unet"]), ("fstext", pf, new["fstext"])):
        for k in ref:
            assert (mine[k].reshape(ref[k].shape) - ref[k]).abs().max() < 1e-5, (name, k)


def test_gradient_accumulation_and_lr_schedule(setup):
    """configs/train.yaml gradient_accumulation_steps: 2 -- the optimizer sees the MEAN of the micro-batch gradients and steps
    on every second call (accelerate's `accumulate` + loss / steps); cosine schedule with warm-up as diffusers defines it."""
    from seervideoldm_amd.trainer import cosine_lr
    usd, fsd, unet, fst = setup
    tr = SeerTrainer(unet, fst, ops=tob, tops=ttob, gradient_accumulation_steps=2, **HP)
    g1u, g2u = torch.randn_like(tr.pu.g), torch.randn_like(tr.pu.g)
    g1f, g2f = torch.randn_like(tr.pf.g), torch.randn_like(tr.pf.g)
    p0 = tr.pu.p.clone()
    tr.pu.g.copy_(g1u); tr.pf.g.copy_(g1f)
    assert tr.accumulate() is False
    tr.pu.g.copy_(g2u); tr.pf.g.copy_(g2f)
    assert tr.accumulate() is True
    assert torch.allclose(tr.pu.acc, (g1u + g2u) / 2) and torch.allclose(tr.pf.acc, (g1f + g2f) / 2)
    tr.optimizer_step()
    ref = SeerTrainer(unet, fst, ops=tob, tops=ttob, **HP)
    ref.pu.g.copy_((g1u + g2u) / 2); ref.pf.g.copy_((g1f + g2f) / 2)
    ref.optimizer_step()
    assert torch.allclose(tr.pu.p, ref.pu.p, atol=1e-7) and torch.allclose(tr.pf.p, ref.pf.p, atol=1e-7)
    assert not torch.equal(tr.pu.p, p0)
    tr.pu.g.copy_(g2u); tr.pf.g.copy_(g2f)
    assert tr.accumulate() is False and torch.allclose(tr.pu.acc, g2u / 2)      # a new window starts from zero
    lr = [cosine_lr(s, 1.0, 10, 110) for s in (0, 5, 10, 60, 110)]
    assert lr[0] == 0.0 and abs(lr[1] - 0.5) < 1e-12 and lr[2] == 1.0 and abs(lr[3] - 0.5) < 1e-12 and abs(lr[4]) < 1e-12


def test_save_state_round_trip(setup, tmp_path):
    """after a step, `save_state` writes the two files inference reads (inference_img.py:98-104) with the UPDATED trainable
    tensors under the reference's names (unpacked q|k|v, de-interleaved GEGLU rows) and every frozen tensor untouched; the
    modules themselves sample with the new weights; the Adam moments come back through load_optimizer_state."""
    from seervideoldm_amd.io import load_seer_checkpoint
    usd, fsd, _, _ = setup
    unet = SeerUNet(**CFG)
    unet.load_state_dict(usd, strict=True)
    fst = FSTextTransformer(num_frames=16, in_channels=192, out_channels=192, n_heads=2, num_layers=1, cross_attention_dim=192)
    fst.load_state_dict(fsd, strict=True)
    fst.set_numframe(3)
    tr = SeerTrainer(unet, fst, ops=tob, tops=ttob, **HP)
    tr.pu.g.normal_(generator=torch.Generator().manual_seed(5)); tr.pf.g.normal_(generator=torch.Generator().manual_seed(6))
    tr.optimizer_step()
    path = tr.save_state(str(tmp_path / "learned_sdunet-steps-1"), global_step=1)
    u2, f2 = SeerUNet(**CFG), FSTextTransformer(num_frames=16, in_channels=192, out_channels=192, n_heads=2, num_layers=1,
                                                cross_attention_dim=192)
    load_seer_checkpoint(path, u2, f2)                                        # strict=True on both
    new = tr.trainable_state_dict()
    sd_u, sd_f = u2.state_dict(), f2.state_dict()
    changed = 0
    for k, v in usd.items():
        if k in new["unet"]:
            assert torch.equal(sd_u[k], new["unet"][k].reshape(v.shape)) and not torch.equal(sd_u[k], v.float())
            changed += 1
        else:
            assert torch.equal(sd_u[k], v), k                                  # frozen tensors and buffers: untouched
    assert changed == len(new["unet"]) > 0
    for k, v in fsd.items():
        if k in new["fstext"]:
            assert torch.equal(sd_f[k], new["fstext"][k].reshape(v.shape)), k
    assert torch.equal(dict(unet.named_parameters())["mid_block.temporal_attentions.0.proj_out.bias"],
                       new["unet"]["mid_block.temporal_attentions.0.proj_out.bias"])
    tr2 = SeerTrainer(u2, f2, ops=tob, tops=ttob, **HP)
    tr2.load_optimizer_state(path)
    assert tr2.step_count == 1 and torch.equal(tr2.pu.m, tr.pu.m) and torch.equal(tr2.pf.v, tr.pf.v)
    assert torch.equal(tr2.pu.p, tr.pu.p) and torch.equal(tr2.pf.p, tr.pf.p)   # repacking the saved files == the live masters


def test_text_loss_option(setup):
    """train.py:346-347 `--text_loss`: the extra loss and its gradient into the FSTextTransformer"""
    usd, fsd, unet, fst = setup
    fst.set_numframe(3)
    tr = SeerTrainer(unet, fst, ops=tob, tops=ttob, text_loss=True, **HP)
    x, noise, text, t = _randn((1, 4, 3, 8, 8), 1), _randn((1, 4, 2, 8, 8), 2), _randn((1, 77, 192), 3), torch.tensor([417])
    loss = tr.forward_backward(x, noise, t, text, 1)
    ref_loss, gu, gf, _ = O.train_loss_and_grads(usd, {**O.DEFAULT_CFG, **CFG}, fsd, x, noise, t, text, 1, fstext_heads=2,
                                                 text_loss=True)
    assert abs(float(loss) - float(ref_loss)) < 2e-2 * float(ref_loss)
    got = tr.trainable_state_dict_of(tr.pu.g, tr.pf.g)["fstext"]
    num = sum(((got[k].reshape(gf[k].shape) - gf[k]) ** 2).sum() for k in gf) ** 0.5
    den = sum((gf[k] ** 2).sum() for k in gf) ** 0.5
    assert num / den < 3e-2, float(num / den)


def test_deferred_weight_gradients_are_final_at_the_hook_and_equal_the_immediate_ones(setup, monkeypatch):
    """the weight gradients (and the LayerNorms' d gamma / d beta finals) of a backward walk wait for its end and go out as one group
    (_lin_bwd / _flush_dw): (1) the same gradients as with one launch per layer (SEER_DW_GROUPED=0); (2) the UNet's gradient segment is
    complete when `on_unet_grads` runs -- a data-parallel step starts its all-reduce there (start_unet_allreduce) -- and the
    FSTextTransformer walk does not touch it afterwards; (3) no queue entry outlives the step"""
    usd, fsd, unet, fst = setup
    B, Fr, cond, H = 1, 3, 1, 8
    fst.set_numframe(Fr)
    x, noise, text = _randn((B, 4, Fr, H, H), 1), _randn((B, 4, Fr - cond, H, H), 2), _randn((B, 77, 192), 3)
    t = torch.tensor([417])
    tr = SeerTrainer(unet, fst, ops=tob, tops=ttob, **HP)
    assert tr._dw_deferred and tr._cf is not None
    seen = {}
    calls = []
    real = ttob.gemm_tn_grouped
    monkeypatch.setattr(ttob, "gemm_tn_grouped", lambda probs: (calls.append(len(probs)), real(probs))[1])
    tr.forward_backward(x, noise, t, text, cond, on_unet_grads=lambda: seen.update(gu=tr.pu.g.clone(), gf=tr.pf.g.clone()))
    assert calls and len(calls) == 2 and min(calls) > 1, calls          # one group per walk (UNet, FSTextTransformer), many layers each
    assert not tr._dw and not tr._cf
    assert torch.equal(seen["gu"], tr.pu.g), "the UNet segment changed after the hook"
    assert not torch.equal(seen["gf"], tr.pf.g)                          # ... while the FSTextTransformer segment was still to come
    monkeypatch.setenv("SEER_DW_GROUPED", "0")
    tr0 = SeerTrainer(unet, fst, ops=tob, tops=ttob, **HP)
    assert not tr0._dw_deferred
    tr0.forward_backward(x, noise, t, text, cond)
    assert torch.equal(tr0.pu.g, tr.pu.g) and torch.equal(tr0.pf.g, tr.pf.g)
    assert len(calls) == 2
