"""Time one fine-tuning step (BASELINE config 5: b=1, F=12, cond_frames=2, 32x32 latent, full-size SeerUNet + 8-layer
FSTextTransformer, random-init weights, synthetic latents) on one MI355X: forward+loss+backward and the optimizer, with HIP
events on the current stream.  Usage: python scripts/bench_train.py [steps] [frames] [latent] [batch] [--eager]"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from seervideoldm_amd import FSTextTransformer, SeerUNet, synth  # noqa: E402
from seervideoldm_amd.trainer import SeerTrainer  # noqa: E402


def build(device, cfg=None, fs_layers=8):
    cfg = cfg or dict(synth.SD15_UNET_CFG)
    unet = SeerUNet(**cfg).to(device)
    unet.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=device), strict=True)
    fst = FSTextTransformer(num_frames=16, num_layers=fs_layers).to(device)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(num_layers=fs_layers), device=device), strict=True)
    return unet, fst


def time_train(device, steps=5, warmup=2, Fr=12, cond=2, lat=32, b=1, use_graph=True, unet=None, fst=None, process_group=None):
    if unet is None or fst is None:
        unet, fst = build(device)
    fst.set_numframe(Fr)
    tr = SeerTrainer(unet, fst, lr=1e-5, max_grad_norm=0.3, process_group=process_group)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((b, 4, Fr, lat, lat), generator=g).to(device)
    noise = torch.randn((b, 4, Fr - cond, lat, lat), generator=g).to(device)
    text = torch.randn((b, 77, 768), generator=g).to(device)
    t = torch.tensor([500] * b, device=device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fb, opt, losses, host = [], [], [], []
    for i in range(warmup + steps):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        ev[0].record()
        loss = tr.forward_backward(x, noise, t, text, cond, use_graph=use_graph,
                                   on_unet_grads=tr.start_unet_allreduce if process_group is not None else None)
        ev[1].record()
        host.append((time.perf_counter() - h0) * 1e3)
        tr.optimizer_step()
        ev[2].record()
        torch.cuda.synchronize()
        losses.append(float(loss))
        if i >= warmup:
            fb.append(ev[0].elapsed_time(ev[1]))
            opt.append(ev[1].elapsed_time(ev[2]))
    n_u, n_f = tr.pu.n, tr.pf.n
    return {"config": f"b={b} F={Fr} cond={cond} latent={lat}x{lat} (BASELINE config 5), bf16 compute, fp32 master/AdamW",
            "hipgraph": bool(use_graph and not getattr(tr, "_graph_broken", False)), "fwd_bwd_ms": sum(fb) / len(fb), "optimizer_ms": sum(opt) / len(opt), "host_enqueue_ms": sum(host[warmup:]) / len(host[warmup:]),
            "ms_per_step": (sum(fb) + sum(opt)) / len(fb), "steps": steps, "trainable_params": int(n_u + n_f),
            "loss_first": losses[0], "loss_last": losses[-1],
            "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}


if __name__ == "__main__":
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(pos[0]) if len(pos) > 0 else 5
    Fr = int(pos[1]) if len(pos) > 1 else 12
    lat = int(pos[2]) if len(pos) > 2 else 32
    b = int(pos[3]) if len(pos) > 3 else 1
    t0 = time.time()
    r = time_train(torch.device("cuda:0"), steps=steps, Fr=Fr, lat=lat, b=b, use_graph="--eager" not in sys.argv)
    r["wall_s"] = time.time() - t0
    print(json.dumps(r))
