"""Hunt for the last-bit replay-vs-eager difference of profiles/r02_colsum_flake.log.

The round-2 observation: two processes on ONE MI355X, each running the width-320 mini network at B_local = 1, 2 frames, 16x16
latent; with GroupNorm statistics from column sums a hipGraph replay differed from the eager run in the last bits in ~7 % of
the test runs.  This script reproduces the setting without torch.distributed (the failing case has no per-layer exchange) and
runs it thousands of times:

    parent                         spawns the workers (never touches the GPU itself)
    worker "main"                  ref = eager run; then alternates eager / replay and compares every output with ref bit
                                   for bit; --trace keeps a clone of every op output and names the FIRST differing op;
                                   --poison fills every torch.empty the host code makes with NaN before the kernels run
    worker "noise" (--cotenant 1)  the same network in a loop on the same GPU, as the second rank was

    python scripts/exp_flake.py --colsums 1 --cotenant 1 --iters 400 [--trace 1] [--poison 1]
"""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CFG_MINI = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=256, attention_head_dim=8)


class _Trace:
    """proxy of seervideoldm_amd.ops that keeps a clone of every tensor an op returns (or writes through out=)"""

    def __init__(self, ops, qkv_tile=0, mode=""):
        self._ops = ops
        self.rec = []
        self.on = False
        self.qkv_tile = qkv_tile
        self.mode = mode

    def __getattr__(self, name):
        f = getattr(self._ops, name)
        if name == "gemm" and (self.qkv_tile or self.mode):
            g, tile, mode = f, self.qkv_tile, self.mode

            def f(a, w, **k):                      # the q|k|v projections on another tile kernel / with another epilogue
                if w.shape[0] == 960 and tile:
                    k["tile"] = tile
                if mode == "nocs":                 # no column scale anywhere
                    k.pop("col_scale", None)
                elif mode == "allcs" and "col_scale" not in k and not k.get("geglu"):
                    k["col_scale"] = (1.0, 64)     # x * 1.0f: same values, but every plain GEMM takes the column-scale pass
                elif mode == "norot":
                    k.pop("rotary", None)
                return g(a, w, **k)
        if not callable(f) or name in ("ColSums", "qk_prescale") or not self.on:
            return f
        import torch

        def wrapped(*a, **k):
            r = f(*a, **k)
            if torch.is_tensor(r):
                self.rec.append((name, tuple(r.shape), r.clone()))
                cs = getattr(r, "colsums", None)
                if cs is not None:
                    self.rec.append((name + ".colsums", tuple(cs.buf.shape), cs.buf.clone()))
            if name == "groupnorm_stats_from_colsums" or name == "groupnorm_stats":
                st = a[4] if len(a) > 4 else k["stats"]
                self.rec.append((name + ".stats", tuple(st.shape), st.clone()))
            return r
        return wrapped


def worker_main(args):
    sys.path.insert(0, str(ROOT))
    import torch
    from seervideoldm_amd import SeerUNet, synth
    from seervideoldm_amd import ops as hip_ops
    from seervideoldm_amd import unet as unet_mod
    dev = torch.device("cuda:0")
    if args.poison:
        real_empty = torch.empty

        def poisoned(*a, **k):
            t = real_empty(*a, **k)
            if t.is_cuda and t.dtype in (torch.float32, torch.bfloat16):
                t.fill_(float("nan"))
            return t
        def poisoned_like(x, **k):
            return poisoned(x.shape, device=x.device, dtype=k.get("dtype", x.dtype))
        hip_ops.torch = type("T", (), {"__getattr__": lambda s, n: poisoned if n == "empty" else
                                       poisoned_like if n == "empty_like" else getattr(torch, n)})()
        unet_mod.torch = hip_ops.torch
    m = SeerUNet(**CFG_MINI).to(dev)
    m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(CFG_MINI), device=dev), strict=True)
    m.eval()
    m.gn_colsums = bool(args.colsums)
    tr = _Trace(hip_ops, args.qkv_tile, args.mode)
    if args.trace or args.qkv_tile or args.mode:
        m._ops_backend = tr
    g = torch.Generator().manual_seed(7)
    B, Fr, H = args.batch, args.frames, args.latent
    x = torch.randn((B, 4, Fr, H, H), generator=g).to(dev)
    ctx = torch.randn((B, Fr, 77, 256), generator=g).to(dev)
    t = torch.tensor([501] * B, device=dev)

    def run(graph):
        m.use_graph = graph
        tr.rec = []
        tr.on = bool(args.trace) and not graph       # eager traces record directly; the replay's clones were made at capture
        out = m(x, t, ctx, cond_frame=0)
        torch.cuda.synchronize()
        return out

    m.use_graph = False
    m(x, t, ctx, cond_frame=0)                     # fills the per-prompt caches (cross-attention K|V, rotary tables)
    tr.on = bool(args.trace)
    tr.rec = []
    ref = m(x, t, ctx, cond_frame=0).clone()
    torch.cuda.synchronize()
    ref_rec = tr.rec
    assert torch.isfinite(ref).all(), "non-finite reference output (poison reached a kernel input?)"
    print(f"[main] colsums={args.colsums} GroupNorms from column sums: {m._engine.gn_from_colsums} of {m._engine.n_groupnorms()}",
          flush=True)
    # capture (with the trace clones inside the graph when tracing)
    tr.rec = []
    tr.on = bool(args.trace)
    m.use_graph = True
    first = m(x, t, ctx, cond_frame=0)
    torch.cuda.synchronize()
    graph_rec = tr.rec if args.trace else []
    # the engine warms up eagerly before capturing: with tracing on, rec holds warm-up clones followed by captured clones
    if args.trace:
        n = len(ref_rec)
        assert len(graph_rec) == 2 * n, (len(graph_rec), n)
        graph_rec = graph_rec[n:]
    bad = {"eager": 0, "replay": 0}
    firsts = {}
    t0 = time.time()
    for it in range(args.iters):
        for mode in ("eager", "replay"):
            out = run(mode == "replay")
            if not torch.equal(out, ref):
                bad[mode] += 1
                nd = int((out != ref).sum())
                rel = float((out - ref).norm() / ref.norm())
                msg = f"[main] it {it} {mode}: {nd} of {out.numel()} outputs differ, rel_l2 {rel:.3g}"
                if args.trace:
                    rec = tr.rec if mode == "eager" else graph_rec
                    for (name, shape, a), (_, _, b) in zip(rec, ref_rec):
                        same = torch.equal(a, b) if not (torch.isnan(a).any() or torch.isnan(b).any()) else \
                            torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
                        if not same:
                            k = (mode, name, shape)
                            firsts[k] = firsts.get(k, 0) + 1
                            idx = [i for i, r in enumerate(rec) if r[2] is a][0]
                            d = (a.float() - b.float()).abs()
                            msg += f"; first differing op #{idx} {name}{shape}: {int((a != b).sum())} elements, max abs {float(d.max()):.3g}"
                            if a.dim() == 2:
                                nz = (a != b).nonzero()
                                rows, cols = sorted(set(nz[:, 0].tolist())), sorted(set(nz[:, 1].tolist()))
                                msg += f"\n        rows {rows} cols {cols}; ops around: " + \
                                    " ".join(f"{r[0]}{r[1]}" for r in rec[max(idx - 3, 0):idx + 3])
                                r0, c0 = int(nz[0, 0]), int(nz[0, 1])
                                msg += f"\n        got  {a[r0, c0:c0 + 8].float().tolist()}\n        want {b[r0, c0:c0 + 8].float().tolist()}"
                            break
                print(msg, flush=True)
    dt = time.time() - t0
    print(f"[main] RESULT colsums={args.colsums} cotenant={args.cotenant} trace={args.trace} poison={args.poison} "
          f"B={B} F={Fr} H={H} iters={args.iters}: eager mismatches {bad['eager']}, replay mismatches {bad['replay']} "
          f"({dt:.1f} s)", flush=True)
    for k, v in firsts.items():
        print(f"[main]   first differing op {k}: {v} times", flush=True)


def worker_noise(args):
    sys.path.insert(0, str(ROOT))
    import torch
    from seervideoldm_amd import SeerUNet, synth
    dev = torch.device("cuda:0")
    m = SeerUNet(**CFG_MINI).to(dev)
    m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(CFG_MINI), device=dev), strict=True)
    m.eval()
    g = torch.Generator().manual_seed(8)
    x = torch.randn((args.batch, 4, args.frames, args.latent, args.latent), generator=g).to(dev)
    ctx = torch.randn((args.batch, args.frames, 77, 256), generator=g).to(dev)
    t = torch.tensor([501] * args.batch, device=dev)
    m.use_graph = bool(args.noise_graph)
    stop = Path(args.stop_file)
    n = 0
    fn = lambda: m(x, t, ctx, cond_frame=0)
    if args.noise_kind == "gemm":
        from seervideoldm_amd import ops
        a = torch.randn((4096, 1280), device=dev).to(torch.bfloat16)
        w = torch.randn((1280, 1280), device=dev).to(torch.bfloat16)
        o = torch.empty((4096, 1280), device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, out=o)
    elif args.noise_kind == "copy":
        a = torch.randn((64 << 20,), device=dev)
        b = torch.empty_like(a)
        fn = lambda: b.copy_(a)
    elif args.noise_kind == "small":
        a = torch.randn((1024,), device=dev)
        fn = lambda: a.add_(1.0)
    elif args.noise_kind in ("qkvgemm", "attn", "ln", "conv", "gn", "wattn"):
        from seervideoldm_amd import ops
        bf = torch.bfloat16
        a = torch.randn((512, 320), device=dev).to(bf)
        w = (torch.randn((960, 320), device=dev) * 0.05).to(bf)
        w9 = (torch.randn((320, 2880), device=dev) * 0.02).to(bf)
        gam, bet = torch.ones(320, device=dev), torch.zeros(320, device=dev)
        qkv = torch.randn((512, 960), device=dev).to(bf)
        att = torch.empty((512, 320), device=dev, dtype=bf)
        stats = torch.empty((1, 32, 2), device=dev)
        kinds = {
            "qkvgemm": lambda: ops.gemm(a, w, col_scale=(0.228, 320)),
            "attn": lambda: ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], att, batch=2, heads=8, head_dim=40, Sq=256,
                                          Sk=256, q_prescaled=True),
            "wattn": lambda: ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], att, batch=1, heads=8, head_dim=40,
                                           Sq=2 * 16, Sk=2 * 16, causal=True, window=(4, 2, 16, 16), q_prescaled=True),
            "ln": lambda: ops.layernorm(a, gam, bet),
            "conv": lambda: ops.conv3x3(a, w9, 2, 16, 16),
            "gn": lambda: ops.groupnorm_apply(a, None, 1, 32, ops.groupnorm_stats(a, None, 1, 32, stats), 512 * 10, 1e-5, gam, bet, True),
        }
        one = kinds[args.noise_kind]
        one()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(50):
                one()
        fn = gr.replay if args.noise_graph else one
    fn()
    torch.cuda.synchronize()
    if args.ready_file:
        Path(args.ready_file).write_text("ready")
    while not stop.exists():
        fn()
        n += 1
        if n % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"[noise] {n} forwards", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--colsums", type=int, default=1)
    ap.add_argument("--cotenant", type=int, default=1)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--trace", type=int, default=0)
    ap.add_argument("--poison", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--latent", type=int, default=16)
    ap.add_argument("--noise-graph", type=int, default=1)
    ap.add_argument("--qkv-tile", type=int, default=0)
    ap.add_argument("--noise-kind", default="model")
    ap.add_argument("--mode", default="")
    ap.add_argument("--ready-file", default="")
    ap.add_argument("--stop-file", default="/tmp/exp_flake.stop")
    args = ap.parse_args()
    if args.role == "main":
        return worker_main(args)
    if args.role == "noise":
        return worker_noise(args)
    stop = Path(args.stop_file)
    if stop.exists():
        stop.unlink()
    base = [sys.executable, __file__] + [a for a in sys.argv[1:]]
    noise = subprocess.Popen(base + ["--role", "noise"]) if args.cotenant else None
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
