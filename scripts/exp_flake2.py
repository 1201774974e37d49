"""Second stage of the flake hunt (scripts/exp_flake.py found the FIRST differing op of a step under co-tenancy: always a
q|k|v projection GEMM, 16 rows x 1 column with column % 16 == 15 -- the last row of one MFMA result tile).

One GEMM launched tens of thousands of times next to a second process on the same GPU; every output is compared bit for bit
with the first.  Cases differ in N, epilogue and tile; co-tenants in what they run.

    python scripts/exp_flake2.py --cotenant model|gemm|copy|spin|none --launches 20000
"""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def worker_main(args):
    import torch
    from seervideoldm_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    bf16 = torch.bfloat16
    cases = []
    for spec in args.cases.split(","):
        name, M, N, K, epi, tile = spec.split(":")
        M, N, K, tile = int(M), int(N), int(K), int(tile)
        a = (torch.randn((M, K), generator=g)).to(dev).to(bf16)
        w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dev).to(bf16)
        bias = torch.randn((N,), generator=g).to(dev)
        res = torch.randn((M, N), generator=g).to(dev).to(bf16)
        kw = dict(tile=tile)
        if "c" in epi:
            kw["col_scale"] = (0.228, min(N, 320))
        if "b" in epi:
            kw["bias"] = bias
        if "r" in epi:
            kw["residual"] = res
        if "R" in epi:          # rotary on the q|k columns, as the temporal q|k|v projection has it
            freqs = (10000.0 ** (-torch.arange(0, 32, 2).float() / 32)).to(dev)
            kw["rotary"] = (ops.rotary_table(freqs, M), M, 0, 40, 32, 640)
        if "g" in epi:
            kw["geglu"] = True
            kw["bias"] = bias
        cases.append((spec, a, w, kw))
    R = 64
    for spec, a, w, kw in cases:
        ref = ops.gemm(a, w, **kw).clone()
        torch.cuda.synchronize()
        outs = [torch.empty_like(ref) for _ in range(R)]
        bad, patterns = 0, {}
        t0 = time.time()
        n = 0
        while n < args.launches or (args.seconds and time.time() - t0 < args.seconds):
            for o in outs:
                ops.gemm(a, w, out=o, **kw)
            n += R
            st = torch.stack(outs)
            ne = (st != ref[None]).flatten(1).any(1)
            if bool(ne.any()):
                for i in ne.nonzero().flatten().tolist():
                    bad += 1
                    nz = (outs[i] != ref).nonzero()
                    rows, cols = sorted(set(nz[:, 0].tolist())), sorted(set(nz[:, 1].tolist()))
                    key = (len(rows), len(cols), tuple(c % 16 for c in cols)[:4], rows[0] % 16)
                    patterns[key] = patterns.get(key, 0) + 1
                    if bad <= 6 and "rotary" in kw and cols[0] % 4 == 3:
                        # which wrong expression is it?  pre-rotation values of the failing quad's second pair from an fp32 product
                        r0, c0 = rows[0], cols[0]
                        pre = a[r0].float() @ w[c0 - 1:c0 + 1].float().t()
                        a1, b1 = float(pre[0]), float(pre[1])
                        tab = kw["rotary"][0]
                        ch = c0 % 40
                        c_, s_ = float(tab[r0, ch // 2, 0]), float(tab[r0, ch // 2, 1])
                        sc = kw["col_scale"][0] if ("col_scale" in kw and c0 < kw["col_scale"][1]) else 1.0
                        cand = {"want b1c+a1s": b1 * c_ + a1 * s_, "a1": a1, "b1": b1, "2*s*a1": 2 * s_ * a1, "a1c-b1s": a1 * c_ - b1 * s_,
                                "s*b1-s*b1": 0.0, "c*b1": c_ * b1, "s*a1": s_ * a1, "2*c*b1": 2 * c_ * b1}
                        print("      candidates (x scale): " + ", ".join(f"{k}={v * sc:.4g}" for k, v in cand.items()), flush=True)
                    if bad <= 6:
                        r0, c0 = rows[0], cols[0]
                        print(f"    launch ~{n}: rows {rows[0]}..{rows[-1]} ({len(rows)}) cols {cols[:6]} got {float(outs[i][r0, c0]):.4g} "
                              f"want {float(ref[r0, c0]):.4g}", flush=True)
        dt = time.time() - t0
        print(f"[main] cotenant={args.cotenant} lib={os.environ.get('SEER_HIP_LIB', 'default')} case {spec}: {bad} bad of {n} launches "
              f"({dt:.1f} s); patterns (rows, cols, col%16, row0%16) {patterns}", flush=True)


def worker_noise(args):
    import torch
    dev = torch.device("cuda:0")
    stop = Path(args.stop_file)
    n = 0
    if args.cotenant == "model":
        from seervideoldm_amd import SeerUNet, synth
        cfg = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=256, attention_head_dim=8)
        m = SeerUNet(**cfg).to(dev)
        m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev), strict=True)
        m.eval()
        m.use_graph = True
        g = torch.Generator().manual_seed(8)
        x = torch.randn((1, 4, 2, 16, 16), generator=g).to(dev)
        ctx = torch.randn((1, 2, 77, 256), generator=g).to(dev)
        t = torch.tensor([501], device=dev)
        fn = lambda: m(x, t, ctx, cond_frame=0)
    elif args.cotenant == "gemm":
        from seervideoldm_amd import ops
        a = torch.randn((4096, 1280), device=dev).to(torch.bfloat16)
        w = torch.randn((1280, 1280), device=dev).to(torch.bfloat16)
        o = torch.empty((4096, 1280), device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, out=o)
    elif args.cotenant == "copy":
        a = torch.randn((64 << 20,), device=dev)
        b = torch.empty_like(a)
        fn = lambda: b.copy_(a)
    elif args.cotenant == "small":
        a = torch.randn((1024,), device=dev)
        fn = lambda: a.add_(1.0)
    else:
        raise SystemExit(f"unknown co-tenant {args.cotenant}")
    while not stop.exists():
        fn()
        n += 1
        if n % 32 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"[noise:{args.cotenant}] {n} iterations", flush=True)


DEFAULT_CASES = ("qkv:512:960:320:c:0,qkv_plain:512:960:320::0,q:512:320:320:c:0,proj_in:512:320:320:b:0,"
                 "qkv_ring:512:960:320:c:8,qkv_128x64:512:960:320:c:3,qkv128:128:960:320:c:0,ff1:512:2560:320:g:0,"
                 "out:512:320:320:br:0,ff2:512:320:1280:br:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--cotenant", default="model")
    ap.add_argument("--launches", type=int, default=20000)
    ap.add_argument("--seconds", type=float, default=0.0)
    ap.add_argument("--cases", default=DEFAULT_CASES)
    ap.add_argument("--stop-file", default="/tmp/exp_flake2.stop")
    args = ap.parse_args()
    if args.role == "main":
        return worker_main(args)
    if args.role == "noise":
        return worker_noise(args)
    stop = Path(args.stop_file)
    if stop.exists():
        stop.unlink()
    base = [sys.executable, __file__] + sys.argv[1:]
    noise = subprocess.Popen(base + ["--role", "noise"]) if args.cotenant != "none" else None
    if noise is not None:
        time.sleep(25)          # let the co-tenant finish importing / capturing before the measurement starts
    rc = subprocess.call(base + ["--role", "main"])
    stop.write_text("stop")
    if noise is not None:
        try:
            noise.wait(timeout=120)
        except subprocess.TimeoutExpired:
            noise.kill()
    stop.unlink()
    sys.exit(rc)


if __name__ == "__main__":
    main()
