"""Third stage of the flake hunt: the q|k|v projection of the FIRST text block with the MODEL'S OWN tensors (LayerNorm input,
weights), isolated from the rest of the step and looped next to a co-tenant process:
    gemm          only the projection
    ln_gemm       LayerNorm -> projection          (the kernel in front of it in the step)
    ln_gemm_attn  LayerNorm -> projection -> self-attention
Every projection output is compared bit for bit with the first.

    python scripts/exp_flake3.py --seconds 20
"""
import argparse
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
CFG_MINI = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=256, attention_head_dim=8)


def worker_main(args):
    import torch
    from seervideoldm_amd import SeerUNet, synth
    from seervideoldm_amd import ops as hip_ops
    dev = torch.device("cuda:0")
    m = SeerUNet(**CFG_MINI).to(dev)
    m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(CFG_MINI), device=dev), strict=True)
    m.eval()
    cap = {}

    class Tap:
        def __getattr__(self, name):
            f = getattr(hip_ops, name)
            if name == "layernorm":
                def g(x, gamma, beta, **k):
                    if "ln" not in cap:
                        cap["ln"] = (x.clone(), gamma, beta)
                    return f(x, gamma, beta, **k)
                return g
            if name == "gemm":
                def g(a, w, **k):
                    if w.shape[0] == 960 and "gemm" not in cap:
                        cap["gemm"] = (a.clone(), w, dict(k))
                    return f(a, w, **k)
                return g
            return f
    m._ops_backend = Tap()
    gen = torch.Generator().manual_seed(7)
    x = torch.randn((1, 4, 2, 16, 16), generator=gen).to(dev)
    ctx = torch.randn((1, 2, 77, 256), generator=gen).to(dev)
    m(x, torch.tensor([501], device=dev), ctx, cond_frame=0)
    torch.cuda.synchronize()
    h, gamma, beta = cap["ln"]
    a, w, kw = cap["gemm"]
    C = 320
    ref = hip_ops.gemm(hip_ops.layernorm(h, gamma, beta), w, **kw).clone()
    assert torch.equal(ref, hip_ops.gemm(a, w, **kw))
    att = torch.empty((a.shape[0], C), device=dev, dtype=torch.bfloat16)

    def seq_gemm():
        return hip_ops.gemm(a, w, **kw)

    def seq_ln_gemm():
        return hip_ops.gemm(hip_ops.layernorm(h, gamma, beta), w, **kw)

    def seq_ln_gemm_attn():
        q = hip_ops.gemm(hip_ops.layernorm(h, gamma, beta), w, **kw)
        hip_ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], att, batch=2, heads=8, head_dim=40, Sq=256, Sk=256, q_prescaled=True)
        return q

    for name, fn in (("gemm", seq_gemm), ("ln_gemm", seq_ln_gemm), ("ln_gemm_attn", seq_ln_gemm_attn)):
        bad, n, t0, pats = 0, 0, time.time(), {}
        while time.time() - t0 < args.seconds:
            outs = [fn() for _ in range(32)]
            n += 32
            st = torch.stack(outs)
            ne = (st != ref[None]).flatten(1).any(1)
            if bool(ne.any()):
                for i in ne.nonzero().flatten().tolist():
                    bad += 1
                    nz = (outs[i] != ref).nonzero()
                    rows, cols = sorted(set(nz[:, 0].tolist())), sorted(set(nz[:, 1].tolist()))
                    key = (len(rows), len(cols), cols[0] % 16)
                    pats[key] = pats.get(key, 0) + 1
        print(f"[main] sequence {name}: {bad} bad of {n} ({time.time() - t0:.1f} s) patterns (rows, cols, col%16) {pats}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--stop-file", default="/tmp/exp_flake.stop")
    args = ap.parse_args()
    if args.role == "main":
        return worker_main(args)
    stop = Path(args.stop_file)
    if stop.exists():
        stop.unlink()
    noise = subprocess.Popen([sys.executable, str(ROOT / "scripts" / "exp_flake.py"), "--role", "noise", "--stop-file", args.stop_file])
    time.sleep(25)
    rc = subprocess.call([sys.executable, __file__, "--role", "main", "--seconds", str(args.seconds)])
    stop.write_text("stop")
    try:
        noise.wait(timeout=120)
    except subprocess.TimeoutExpired:
        noise.kill()
    stop.unlink()
    sys.exit(rc)


if __name__ == "__main__":
    main()
