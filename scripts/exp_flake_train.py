"""The co-tenant check of scripts/exp_flake.py for the FINE-TUNING step: a second process replays the mini denoising network
on the same GPU while this one runs forward + backward of a reduced (real channel widths) training step over and over on
the same micro-batch; loss and the flat gradient buffers must be bit-identical every time (no optimizer step in between).

    python scripts/exp_flake_train.py --iters 300
"""
import argparse
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def worker(args):
    import torch
    from seervideoldm_amd import FSTextTransformer, SeerUNet, synth
    from seervideoldm_amd.trainer import SeerTrainer
    dev = torch.device("cuda:0")
    cfg = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
    fs = dict(num_frames=16, num_layers=1, channels=192, n_heads=2, cross_attention_dim=192)
    unet = SeerUNet(**cfg)
    unet.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev), strict=True)
    fst = FSTextTransformer(num_frames=16, in_channels=192, out_channels=192, n_heads=2, num_layers=1, cross_attention_dim=192)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**fs), device=dev), strict=True)
    fst.set_numframe(4)
    tr = SeerTrainer(unet.to(dev), fst.to(dev), lr=1e-5, max_grad_norm=0.3)
    g = torch.Generator().manual_seed(1)
    x, noise = torch.randn((1, 4, 4, 32, 32), generator=g).to(dev), torch.randn((1, 4, 3, 32, 32), generator=g).to(dev)
    text, t = torch.randn((1, 77, 192), generator=g).to(dev), torch.tensor([417], device=dev)
    for use_graph in (False, True):
        ref = None
        bad = 0
        t0 = time.time()
        for it in range(args.iters):
            tr.pu.g.zero_(); tr.pf.g.zero_()
            loss = tr.forward_backward(x, noise, t, text, 1, use_graph=use_graph)
            torch.cuda.synchronize()
            cur = (float(loss), tr.pu.g.clone(), tr.pf.g.clone())
            if ref is None:
                ref = cur
            elif cur[0] != ref[0] or not torch.equal(cur[1], ref[1]) or not torch.equal(cur[2], ref[2]):
                bad += 1
                if bad <= 3:
                    du = int((cur[1] != ref[1]).sum()); df = int((cur[2] != ref[2]).sum())
                    print(f"   it {it}: loss {cur[0]!r} vs {ref[0]!r}, {du} unet / {df} fstext gradient words differ", flush=True)
        print(f"[train] cotenant={args.cotenant} graph={use_graph}: {bad} of {args.iters - 1} repeats differ from the first "
              f"({time.time() - t0:.1f} s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--cotenant", type=int, default=1)
    ap.add_argument("--stop-file", default="/tmp/exp_flake_train.stop")
    args = ap.parse_args()
    if args.role == "main":
        return worker(args)
    stop, ready = Path(args.stop_file), Path(args.stop_file + ".ready")
    for f in (stop, ready):
        if f.exists():
            f.unlink()
    noise = None
    if args.cotenant:
        noise = subprocess.Popen([sys.executable, str(ROOT / "scripts" / "exp_flake.py"), "--role", "noise", "--stop-file", str(stop),
                                  "--ready-file", str(ready)])
        t0 = time.time()
        while not ready.exists() and time.time() - t0 < 300:
            time.sleep(0.5)
    rc = subprocess.call([sys.executable, __file__, "--role", "main", "--iters", str(args.iters), "--cotenant", str(args.cotenant)])
    stop.write_text("stop")
    if noise is not None:
        try:
            noise.wait(timeout=120)
        except subprocess.TimeoutExpired:
            noise.kill()
    sys.exit(rc)


if __name__ == "__main__":
    main()
