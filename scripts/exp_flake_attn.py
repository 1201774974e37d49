"""Co-tenant check of the d = 40 attention at the step's shape ([192, 1024, 40]: the 64-queries-per-wave form) and of the windowed
causal temporal block: a second process replays the mini network on the same GPU while this one launches the kernel over and over
on the same operands; every output must be bit-identical to the first.

    python scripts/exp_flake_attn.py --seconds 20
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
    from seervideoldm_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    bf16 = torch.bfloat16
    cases = [("spatial [192,1024,40] (64 queries per wave)", dict(batch=24, heads=8, head_dim=40, Sq=1024, Sk=1024), 24 * 1024),
             ("temporal window block (32 queries per wave)", dict(batch=2, heads=8, head_dim=40, Sq=12 * 64, Sk=12 * 64, causal=True,
                                                               window=(8, 12, 32, 32)), 2 * 12 * 1024)]
    for name, kw, rows in cases:
        qkv = torch.randn((rows, 960), generator=g).to(dev).to(bf16)
        ref = torch.zeros((rows, 320), device=dev, dtype=bf16)
        ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], ref, **kw)
        outs = [torch.zeros_like(ref) for _ in range(16)]
        bad = n = 0
        t0 = time.time()
        while time.time() - t0 < args.seconds:
            for o in outs:
                ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], o, **kw)
            n += len(outs)
            bad += int((torch.stack(outs) != ref[None]).flatten(1).any(1).sum())
        print(f"[attn] {name}: {bad} of {n} launches differ from the first", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--stop-file", default="/tmp/exp_flake_attn.stop")
    args = ap.parse_args()
    if args.role == "main":
        return worker(args)
    stop, ready = Path(args.stop_file), Path(args.stop_file + ".ready")
    for f in (stop, ready):
        if f.exists():
            f.unlink()
    noise = subprocess.Popen([sys.executable, str(ROOT / "scripts" / "exp_flake.py"), "--role", "noise", "--stop-file", str(stop),
                              "--ready-file", str(ready)])
    t0 = time.time()
    while not ready.exists() and time.time() - t0 < 300:
        time.sleep(0.5)
    rc = subprocess.call([sys.executable, __file__, "--role", "main", "--seconds", str(args.seconds)])
    stop.write_text("stop")
    try:
        noise.wait(timeout=120)
    except subprocess.TimeoutExpired:
        noise.kill()
    sys.exit(rc)


if __name__ == "__main__":
    main()
