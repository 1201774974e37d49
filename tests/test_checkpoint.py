"""Checkpoint wire format and resume (train.py:68-110, 268-280, 395-399) on the CPU stand-in kernels: a run interrupted after
`save_checkpoint` and continued through `load_checkpoint` in a FRESH trainer must equal the uninterrupted run bit for bit,
the files must be the ones accelerate / train.py name, and optimizer.bin must load into a real torch.optim.AdamW."""
import os

import pytest
import torch

from oracle import ref_import
from seervideoldm_amd import FSTextTransformer, SeerUNet, synth
from seervideoldm_amd.checkpoint import (RunningAverageMeter, load_checkpoint, optimizer_state_dict, reference_param_order,
                                         save_checkpoint)
from seervideoldm_amd.trainer import SeerTrainer, cosine_lr
from tests import torch_ops_backend as tob
from tests import torch_train_ops_backend as ttob

CFG = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
FS = dict(num_frames=16, num_layers=1, channels=192, n_heads=2, cross_attention_dim=192)
HP = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=0.3)
SCHED = dict(base_lr=1e-3, warmup_steps=2, total_steps=10)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _fresh():
    unet = SeerUNet(**CFG)
    unet.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(CFG)), strict=True)
    fst = FSTextTransformer(num_frames=FS["num_frames"], in_channels=192, out_channels=192, n_heads=2, num_layers=1,
                            cross_attention_dim=192)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**FS)), strict=True)
    fst.set_numframe(3)
    unet._ops_backend = tob
    fst._ops_backend = tob
    return SeerTrainer(unet, fst, ops=tob, tops=ttob, **HP)


def _batch(step):
    x0, lat = _randn((1, 4, 1, 8, 8), 10 + step), _randn((1, 4, 2, 8, 8), 20 + step)
    noise, text = _randn((1, 4, 2, 8, 8), 30 + step), _randn((1, 77, 192), 40 + step)
    return x0, lat, noise, torch.tensor([100 + 97 * step]), text


def _run(tr, steps, lr_meter, losses, start=0):
    acp = torch.cumprod(1.0 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2, 0)
    for s in range(start, start + steps):
        lr = cosine_lr(s, **SCHED)
        loss = tr.train_step(*_batch(s)[:4], _batch(s)[4], acp, lr=lr)
        losses.update(float(loss), s + 1)
        lr_meter.update(lr, s + 1)


def test_resume_equals_the_uninterrupted_run(tmp_path):
    out = str(tmp_path)
    # uninterrupted: 4 optimizer steps
    a = _fresh()
    lm_a, ls_a = RunningAverageMeter(), RunningAverageMeter()
    _run(a, 4, lm_a, ls_a)
    # interrupted after 2, saved, resumed in a fresh trainer
    b = _fresh()
    lm_b, ls_b = RunningAverageMeter(), RunningAverageMeter()
    _run(b, 2, lm_b, ls_b)
    path, side = save_checkpoint(b, out, global_step=2, epoch=0, lr_meter=lm_b, losses_train=ls_b, lr=cosine_lr(1, **SCHED),
                                 schedule=SCHED)
    assert os.path.basename(path) == "learned_sdunet-steps-2" and os.path.basename(side) == "learned_sdunet-steps-2.pt"
    assert sorted(os.listdir(path)) == ["optimizer.bin", "pytorch_model.bin", "pytorch_model_1.bin", "random_states_0.pkl",
                                        "scheduler.bin"]
    # accelerate's save_state / load_state write and read the random states with torch.save / torch.load: the file must be
    # a torch archive (a plain pickle is rejected there with "Invalid magic number"), holding accelerate's four keys
    rng = torch.load(os.path.join(path, "random_states_0.pkl"), map_location="cpu", weights_only=False)
    assert {"random_state", "numpy_random_seed", "torch_manual_seed"} <= set(rng)
    sd = torch.load(side, weights_only=False)
    assert set(sd) == {"epoch", "global_step", "lr_meter", "losses_train"} and sd["global_step"] == 2
    assert set(sd["losses_train"]) == {"vals", "avg", "steps"} and sd["losses_train"]["steps"] == [1, 2]
    c = _fresh()
    lm_c, ls_c = RunningAverageMeter(), RunningAverageMeter()
    assert load_checkpoint(c, out, 7, lm_c, ls_c) is None            # nothing saved under that step: start from scratch
    st = load_checkpoint(c, out, 2, lm_c, ls_c)
    assert st == {"global_step": 2, "epoch": 0} and c.step_count == 2
    assert ls_c.vals == ls_b.vals and ls_c.avg == ls_b.avg and lm_c.steps == [1, 2]
    _run(c, 2, lm_c, ls_c, start=st["global_step"])
    for P_a, P_c in ((a.pu, c.pu), (a.pf, c.pf)):
        assert torch.equal(P_a.p, P_c.p) and torch.equal(P_a.m, P_c.m) and torch.equal(P_a.v, P_c.v)
    assert ls_c.vals == ls_a.vals and abs(ls_c.avg - ls_a.avg) < 1e-12


def test_random_states_round_trip_and_tolerate_a_foreign_file(tmp_path):
    """the generator states come back through torch.load; a file this process cannot use is skipped with a warning"""
    import random
    import warnings
    out = str(tmp_path)
    tr = _fresh()
    lm, ls = RunningAverageMeter(), RunningAverageMeter()
    torch.manual_seed(1234)
    random.seed(99)
    path, _ = save_checkpoint(tr, out, global_step=1, epoch=0, lr_meter=lm, losses_train=ls)
    want_t, want_r = torch.rand(3), random.random()
    torch.manual_seed(1)
    random.seed(1)
    load_checkpoint(tr, out, 1, lm, ls)
    assert torch.equal(torch.rand(3), want_t) and random.random() == want_r
    with open(os.path.join(path, "random_states_0.pkl"), "wb") as f:
        f.write(b"not a torch archive")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert load_checkpoint(tr, out, 1, lm, ls) is not None
    assert any("random states" in str(x.message) for x in w)


def test_optimizer_bin_loads_into_torch_adamw(tmp_path):
    """optimizer.bin is `torch.optim.AdamW.state_dict()` over the parameters in train.py:213's order"""
    tr = _fresh()
    _run(tr, 1, RunningAverageMeter(), RunningAverageMeter())
    sd = optimizer_state_dict(tr)
    un = [(k, p) for k, p in tr.unet.named_parameters() if ".temporal_attentions." in k]
    fn = list(tr.fstext.named_parameters())
    order = reference_param_order([k for k, _ in un], [k for k, _ in fn])
    assert sd["param_names"] == order
    lookup = {**dict(un), **{k: p for k, p in fn}}
    params = [torch.nn.Parameter(torch.zeros_like(lookup[k])) for k in order]
    opt = torch.optim.AdamW(params, lr=HP["lr"], betas=HP["betas"], weight_decay=HP["weight_decay"], eps=HP["eps"])
    opt.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
    got = tr.trainable_state_dict_of(tr.pu.m, tr.pf.m)
    k0 = order[0]
    assert torch.equal(opt.state[params[0]]["exp_avg"], got["unet"][k0].reshape(params[0].shape).cpu())
    assert float(opt.state[params[-1]]["step"]) == 1.0


@pytest.mark.skipif(not ref_import.available(), reason="needs /root/reference (build container)")
def test_parameter_order_is_the_references():
    """train.py:213: filter(requires_grad, sunet.parameters()) + fstext_model.parameters() on the REAL modules"""
    ref = ref_import.load_reference()
    cfg = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(32, 64, 64, 64), cross_attention_dim=64,
               attention_head_dim=8, layers_per_block=2)
    r_un = [k for k, _ in ref.unet.SeerUNet(**cfg).named_parameters() if ".temporal_attentions." in k]
    o_un = [k for k, _ in SeerUNet(**cfg).named_parameters() if ".temporal_attentions." in k]
    r_fn = [k for k, _ in ref.unet.FSTextTransformer(num_frames=6, in_channels=192, out_channels=192, n_heads=2, num_layers=2,
                                                      cross_attention_dim=192).named_parameters()]
    o_fn = [k for k, _ in FSTextTransformer(num_frames=6, in_channels=192, out_channels=192, n_heads=2, num_layers=2,
                                            cross_attention_dim=192).named_parameters()]
    assert reference_param_order(o_un, o_fn) == r_un + r_fn


# ---- a directory written by the REAL accelerate.Accelerator.save_state around the REAL reference modules ------------------------------
# tests/golden/accelerate_state/ (oracle/make_goldens_accel.py: two optimizer steps of train.py:343-387 under accelerate 1.14, then
# train.py:395-399's save_state + sidecar).  The product must resume from it: weights bit for bit, both Adam moments in the reference's
# parameter order (the file carries NO names), the step count, the meters.
ACC = os.path.join(os.path.dirname(__file__), "golden", "accelerate_state")
CFG_ACC = dict(block_out_channels=(32, 32, 32, 32), layers_per_block=1, cross_attention_dim=40, attention_head_dim=8)
FS_ACC = dict(num_frames=4, num_layers=1, channels=40, n_heads=1, cross_attention_dim=40)


def _fp(t):
    import numpy as np
    f = t.detach().double().flatten()
    return np.asarray([f.sum().item(), (f * f).sum().item(), f[0].item(), f[-1].item()])


def _fresh_acc():
    unet = SeerUNet(**CFG_ACC)
    fst = FSTextTransformer(num_frames=FS_ACC["num_frames"], in_channels=40, out_channels=40, n_heads=1, num_layers=1, cross_attention_dim=40)
    fst.set_numframe(3)
    unet._ops_backend = tob
    fst._ops_backend = tob
    return SeerTrainer(unet, fst, ops=tob, tops=ttob, **HP)


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
def test_resume_from_a_real_accelerate_save_state_directory(tmp_path, fmt):
    import shutil

    import numpy as np
    exp = np.load(os.path.join(ACC, "expected.npz"))
    work = str(tmp_path / "run")
    shutil.copytree(ACC, work)
    d = os.path.join(work, "learned_sdunet-steps-2")
    assert sorted(os.listdir(d)) == ["optimizer.bin", "pytorch_model.bin", "pytorch_model_1.bin", "random_states_0.pkl", "scheduler.bin"]
    if fmt == "safetensors":
        # today's accelerate default (safe_serialization=True) names the same tensors model.safetensors / model_1.safetensors; the
        # committed directory is the .bin form of the accelerate the reference pins, re-serialised here to exercise that branch
        from safetensors.torch import save_file
        for old, new in (("pytorch_model.bin", "model.safetensors"), ("pytorch_model_1.bin", "model_1.safetensors")):
            sd = torch.load(os.path.join(d, old), map_location="cpu")
            save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, new))
            os.remove(os.path.join(d, old))
    tr = _fresh_acc()
    lm, ls = RunningAverageMeter(), RunningAverageMeter()
    st = load_checkpoint(tr, work, 2, lm, ls)
    assert st == {"global_step": 2, "epoch": 0}
    assert ls.vals == list(exp["loss_vals"]) and ls.steps == [0, 1] and lm.vals == list(exp["lr_vals"])
    # weights: every tensor of both models, bit for bit
    usd, fsd = tr.unet.state_dict(), tr.fstext.state_dict()
    assert set(usd) == {str(k) for k in exp["unet_keys"]} and set(fsd) == {str(k) for k in exp["fstext_keys"]}      # (by name: load order is free)
    for i, k in enumerate(exp["unet_keys"]):
        assert np.array_equal(_fp(usd[str(k)]), exp["unet_fp"][i]), k
    for i, k in enumerate(exp["fstext_keys"]):
        assert np.array_equal(_fp(fsd[str(k)]), exp["fstext_fp"][i]), k
    # optimizer: the file has no names -- the product assumes train.py:213's order, which must be the order the real modules gave
    un = [k for k, _ in tr.unet.named_parameters() if ".temporal_attentions." in k]
    fn = [k for k, _ in tr.fstext.named_parameters()]
    order = reference_param_order(un, fn)
    want = [str(k) for k in exp["param_names"]]
    assert [k if i < len(un) else "fstext:" + k for i, k in enumerate(order)] == want
    assert tr.step_count == int(exp["opt_step"]) == 2
    back = optimizer_state_dict(tr)
    for i in range(len(order)):
        assert np.array_equal(_fp(back["state"][i]["exp_avg"]), exp["exp_avg_fp"][i]), want[i]
        assert np.array_equal(_fp(back["state"][i]["exp_avg_sq"]), exp["exp_avg_sq_fp"][i]), want[i]
    # scheduler.bin is a LambdaLR state: the product recomputes the rate from the step (trainer.cosine_lr) -- same number
    sch = torch.load(os.path.join(d, "scheduler.bin"), map_location="cpu", weights_only=False)
    assert sch["last_epoch"] == 2 and abs(sch["_last_lr"][0] - float(exp["last_lr"])) < 1e-15
    assert abs(sch["_last_lr"][0] - cosine_lr(2, HP["lr"], 1, 10)) < 1e-12
    # (stepping on after a resume is test_resume_equals_the_uninterrupted_run's job, at a width the kernels take: K % 64 == 0)
