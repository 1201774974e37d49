"""Training checkpoints in the reference's wire format, and resume (train.py:68-110, 268-280, 395-399).

train.py saves, every `save_steps` optimizer steps,

    <output_dir>/learned_sdunet-steps-N/         accelerator.save_state(): pytorch_model.bin (SeerUNet), pytorch_model_1.bin
                                                 (FSTextTransformer), optimizer.bin (torch.optim.AdamW.state_dict()),
                                                 scheduler.bin, random_states_{rank}.pkl (torch.save, as accelerate writes it)
    <output_dir>/learned_sdunet-steps-N.pt       {"epoch", "global_step", "lr_meter", "losses_train"}  (the sidecar)

and, on start, loads `learned_sdunet-steps-{saved_global_step}` + its sidecar when they exist.  `save_checkpoint` /
`load_checkpoint` write and read exactly those files for a `SeerTrainer`: the model files are the state dicts inference reads
back (inference_img.py:98-104); optimizer.bin is a torch AdamW state dict whose parameter indices follow the REFERENCE's
parameter order (`filter(requires_grad, sunet.parameters()) + fstext_model.parameters()`, train.py:213 -- the reference
registers up_blocks before mid_block) with the names alongside, so either side can load the other's file.
Host-side only: no kernels.
"""
from __future__ import annotations

import os
import random
from typing import Dict, List, Optional

import numpy as np
import torch

from .trainer import SeerTrainer, _pack_fstext_fp32, _pack_temporal_fp32, cosine_lr


class RunningAverageMeter:
    """train.py:68-110: exponential running average with the value / step history that the sidecar stores"""

    def __init__(self, momentum: float = 0.99, save_seq: bool = True):
        self.momentum, self.save_seq = momentum, save_seq
        self.vals: List[float] = []
        self.steps: List[int] = []
        self.val, self.avg = None, 0

    def reset(self):
        self.val, self.avg = None, 0

    def update(self, val, step=None):
        self.avg = val if self.val is None else self.avg * self.momentum + val * (1 - self.momentum)
        self.val = val
        if self.save_seq:
            self.vals.append(val)
            if step is not None:
                self.steps.append(step)

    def synchronize_and_update(self, val, step=None, process_group=None):
        """mean of `val` over the ranks, then update (train.py:100-110 reduces through the accelerator)"""
        import torch.distributed as dist
        t = torch.as_tensor(val, dtype=torch.float32).detach().clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            dist.all_reduce(t, group=process_group)
            t /= dist.get_world_size(process_group)
        self.update(float(t), step)

    def ckpt(self):
        return {"vals": self.vals, "avg": self.avg, "steps": self.steps}

    def load(self, d):
        self.vals = d["vals"]
        if len(self.vals) > 0:
            self.val = self.vals[-1]
        self.avg, self.steps = d["avg"], d["steps"]


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def reference_param_order(unet_trainable: List[str], fstext_names: List[str]) -> List[str]:
    """names in the order of train.py:213's parameter list.  SeerUNet registers down_blocks, up_blocks, mid_block in that order
    (unet_3d_condition.py:124-203); inside a block our module tree keeps the reference's order."""
    rank = lambda k: 0 if k.startswith("down_blocks.") else (1 if k.startswith("up_blocks.") else 2)
    return sorted(unet_trainable, key=rank) + list(fstext_names)       # sorted() is stable


def _names(tr: SeerTrainer):
    un = [k for k, _ in tr.unet.named_parameters() if ".temporal_attentions." in k]
    fn = [k for k, _ in tr.fstext.named_parameters()]
    return un, fn


def optimizer_state_dict(tr: SeerTrainer, lr: Optional[float] = None) -> Dict:
    """the trainer's Adam moments as `torch.optim.AdamW.state_dict()` in the reference's parameter order"""
    un, fn = _names(tr)
    order = reference_param_order(un, fn)
    m = tr.trainable_state_dict_of(tr.pu.m, tr.pf.m)
    v = tr.trainable_state_dict_of(tr.pu.v, tr.pf.v)
    flat_m = {**m["unet"], **{("fstext:" + k): t for k, t in m["fstext"].items()}}
    flat_v = {**v["unet"], **{("fstext:" + k): t for k, t in v["fstext"].items()}}
    key = lambda k, i: k if i < len(un) else "fstext:" + k
    params_u = dict(tr.unet.named_parameters())
    params_f = dict(tr.fstext.named_parameters())
    state = {}
    for i, k in enumerate(order):
        shape = (params_u[k] if i < len(un) else params_f[k]).shape
        state[i] = {"step": torch.tensor(float(tr.step_count)),
                    "exp_avg": flat_m[key(k, i)].reshape(shape).cpu(),
                    "exp_avg_sq": flat_v[key(k, i)].reshape(shape).cpu()}
    group = {"lr": tr.lr if lr is None else lr, "betas": tuple(tr.betas), "eps": tr.eps, "weight_decay": tr.weight_decay,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "params": list(range(len(order)))}
    return {"state": state, "param_groups": [group], "param_names": order, "n_unet": len(un), "micro": tr._micro}


def load_optimizer_state_dict(tr: SeerTrainer, sd: Dict) -> None:
    """inverse of optimizer_state_dict; also accepts a file written by the reference (no names: its own order is assumed)"""
    un, fn = _names(tr)
    order = sd.get("param_names") or reference_param_order(un, fn)
    n_u = sd.get("n_unet", len(un))
    assert len(order) == len(sd["state"]) == len(un) + len(fn), "optimizer.bin does not match the trainable parameters"
    for which, P, packer in (("exp_avg", "m", None), ("exp_avg_sq", "v", None)):
        du = {order[i]: sd["state"][i][which].float() for i in range(n_u)}
        df = {order[i]: sd["state"][i][which].float() for i in range(n_u, len(order))}
        # frozen tensors are not in the file: the packers only touch the trainable names
        pu = _pack_temporal_fp32(du)
        pf = _pack_fstext_fp32(df, tr.fstext.num_layers)
        for params, packed in ((tr.pu, pu), (tr.pf, pf)):
            flat = getattr(params, P)
            for k in params.names:
                params.view(flat, k).copy_(packed[k].reshape(params.shapes[k]).to(flat.device))
    steps = [int(s["step"]) for s in sd["state"].values()]
    tr.step_count = steps[0] if steps else 0
    tr._micro = int(sd.get("micro", 0))


def save_checkpoint(tr: SeerTrainer, output_dir: str, global_step: int, epoch: int, lr_meter: RunningAverageMeter,
                    losses_train: RunningAverageMeter, *, lr: Optional[float] = None, schedule: Optional[Dict] = None):
    """train.py:395-399.  `schedule` = dict(base_lr, warmup_steps, total_steps) of the cosine schedule in use (scheduler.bin)."""
    save_path = os.path.join(output_dir, f"learned_sdunet-steps-{global_step}")
    os.makedirs(save_path, exist_ok=True)
    tr.sync_modules()
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}
    torch.save(cpu(tr.unet.state_dict()), os.path.join(save_path, "pytorch_model.bin"))
    torch.save(cpu(tr.fstext.state_dict()), os.path.join(save_path, "pytorch_model_1.bin"))
    torch.save(optimizer_state_dict(tr, lr), os.path.join(save_path, "optimizer.bin"))
    sch = dict(schedule or {})
    base = sch.get("base_lr", tr.lr)
    last = cosine_lr(global_step, base, sch.get("warmup_steps", 0), sch.get("total_steps", max(global_step, 1))) if schedule else base
    torch.save({"base_lrs": [base], "last_epoch": global_step, "_step_count": global_step + 1, "_last_lr": [last],
                "lr_lambdas": [None], **{k: sch[k] for k in ("warmup_steps", "total_steps") if k in sch}},
               os.path.join(save_path, "scheduler.bin"))
    rng = {"random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
           "torch_manual_seed": torch.get_rng_state()}
    if torch.cuda.is_available():
        rng["torch_cuda_manual_seed"] = torch.cuda.get_rng_state_all()
    # accelerate's save_state writes this file with torch.save, one per process (random_states_{process_index}.pkl)
    torch.save(rng, os.path.join(save_path, f"random_states_{_rank()}.pkl"))
    side = os.path.join(output_dir, f"learned_sdunet-steps-{global_step}.pt")
    torch.save({"epoch": epoch, "global_step": global_step, "lr_meter": lr_meter.ckpt(), "losses_train": losses_train.ckpt()}, side)
    return save_path, side


def _load_model_file(load_path: str, index: int) -> Dict[str, torch.Tensor]:
    """the `index`-th prepared model of accelerator.save_state: `pytorch_model[_i].bin` (the accelerate the reference pins, and
    `safe_serialization=False` today) or `model[_i].safetensors` (accelerate >= 0.22's default)"""
    suffix = "" if index == 0 else f"_{index}"
    fp = os.path.join(load_path, f"pytorch_model{suffix}.bin")
    if os.path.exists(fp):
        return torch.load(fp, map_location="cpu")
    fs = os.path.join(load_path, f"model{suffix}.safetensors")
    if os.path.exists(fs):
        from safetensors.torch import load_file
        return load_file(fs, device="cpu")
    raise FileNotFoundError(f"neither {fp} nor {fs}: not an accelerator.save_state directory")


def load_checkpoint(tr: SeerTrainer, output_dir: str, saved_global_step: int, lr_meter: RunningAverageMeter,
                    losses_train: RunningAverageMeter, restore_rng: bool = True) -> Optional[Dict]:
    """train.py:268-280: load `learned_sdunet-steps-{saved_global_step}` and its sidecar if they exist.  Returns
    {"global_step", "epoch"} from the sidecar, or None when there is nothing to resume from."""
    load_path = os.path.join(output_dir, f"learned_sdunet-steps-{saved_global_step}")
    side = load_path + ".pt"
    out = None
    if os.path.exists(load_path):
        tr.unet.load_state_dict(_load_model_file(load_path, 0), strict=True)
        tr.fstext.load_state_dict(_load_model_file(load_path, 1), strict=True)
        tr.reload_from_modules()
        load_optimizer_state_dict(tr, torch.load(os.path.join(load_path, "optimizer.bin"), map_location="cpu", weights_only=False))
        rp = os.path.join(load_path, f"random_states_{_rank()}.pkl")
        if restore_rng and os.path.exists(rp):
            # as accelerate's load_state does: a generator state that does not fit this process (another device count, another
            # library version) is skipped with a warning instead of failing the resume
            try:
                rng = torch.load(rp, map_location="cpu", weights_only=False)
                random.setstate(rng["random_state"])
                np.random.set_state(rng["numpy_random_seed"])
                torch.set_rng_state(rng["torch_manual_seed"])
                if torch.cuda.is_available() and "torch_cuda_manual_seed" in rng:
                    torch.cuda.set_rng_state_all(rng["torch_cuda_manual_seed"])
            except Exception as e:      # noqa: BLE001
                import warnings
                warnings.warn(f"could not restore the random states from {rp}: {type(e).__name__}: {e}")
    if os.path.exists(side):
        st = torch.load(side, map_location="cpu", weights_only=False)
        lr_meter.load(st["lr_meter"])
        losses_train.load(st["losses_train"])
        out = {"global_step": st["global_step"], "epoch": st["epoch"]}
    return out
