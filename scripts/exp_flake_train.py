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


class _Trace:
    """proxy of an ops module: keeps a clone of every tensor an op returns or is told to write (out=, dx=, dgamma=, dbeta=,
    lse=, colsum=; attention's out and attention_bwd's dq, dk, dv), so the FIRST differing op of a bad repeat can be named"""
    OUT_KW = ("out", "dx", "dgamma", "dbeta", "lse", "colsum", "dres_out")
    OUT_POS = {"attention": (3,), "attention_bwd": (6, 7, 8), "rotary_inplace": (0,), "groupnorm_stats": (4,)}

    def __init__(self, mod, tag, rec):
        self._mod, self._tag, self.rec, self.on = mod, tag, rec, False

    def __getattr__(self, name):
        f = getattr(self._mod, name)
        if not callable(f) or isinstance(f, type) or name.startswith("_") or name in ("attn_lse_buffer",):
            return f
        import torch

        def wrapped(*a, **k):
            r = f(*a, **k)
            if not self.on[0]:
                return r
            outs = []
            for t in (r if isinstance(r, (tuple, list)) else (r,)):
                if torch.is_tensor(t):
                    outs.append(("ret", t))
            for kw in self.OUT_KW:
                if torch.is_tensor(k.get(kw)):
                    outs.append((kw, k[kw]))
            for i in self.OUT_POS.get(name, ()):
                if i < len(a) and torch.is_tensor(a[i]):
                    outs.append((f"arg{i}", a[i]))
            for what, t in outs:
                self.rec.append((f"{self._tag}.{name}:{what}", tuple(t.shape), t.detach().clone()))
            return r
        return wrapped


def _describe(name, shape, a, b):
    import torch
    ne = a != b
    msg = f"{name}{shape} {a.dtype}: {int(ne.sum())} elements differ, max abs {float((a.float() - b.float()).abs().max()):.3g}"
    if a.dim() == 2:
        nz = ne.nonzero()
        rows, cols = sorted(set(nz[:, 0].tolist())), sorted(set(nz[:, 1].tolist()))
        msg += f"\n        rows({len(rows)}) {rows[:40]}\n        cols({len(cols)}) {cols[:80]}"
        r0, c0 = int(nz[0, 0]), int(nz[0, 1])
        msg += f"\n        got  {a[r0, c0:c0 + 8].float().tolist()}\n        want {b[r0, c0:c0 + 8].float().tolist()}"
    else:
        nz = ne.reshape(-1).nonzero().reshape(-1)
        msg += f"\n        flat idx {nz[:40].tolist()}"
    return msg


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
    rec, on = [], [False]
    if args.trace:
        tr.ops, tr.tops = _Trace(tr.ops, "ops", rec), _Trace(tr.tops, "tops", rec)
        tr.ops.on = tr.tops.on = on
        tr.forward_backward(x, noise, t, text, 1, use_graph=False)          # fills the caches (transposed weights, tables)
        torch.cuda.synchronize()
    ref_rec, firsts = None, {}
    for use_graph in ((False,) if args.trace else (True, False) if args.graph_first else (False, True)):
        ref = None
        bad = 0
        t0 = time.time()
        for it in range(args.iters):
            tr.pu.g.zero_(); tr.pf.g.zero_()
            del rec[:]
            on[0] = bool(args.trace)
            loss = tr.forward_backward(x, noise, t, text, 1, use_graph=use_graph)
            on[0] = False
            torch.cuda.synchronize()
            if args.trace and ref_rec is None:
                ref_rec = list(rec)
                print(f"[train] trace: {len(ref_rec)} op outputs per step, "
                      f"{sum(r[2].numel() * r[2].element_size() for r in ref_rec) / 2**30:.2f} GiB", flush=True)
            cur = (float(loss), tr.pu.g.clone(), tr.pf.g.clone())
            if ref is None:
                ref = cur
            elif cur[0] != ref[0] or not torch.equal(cur[1], ref[1]) or not torch.equal(cur[2], ref[2]):
                bad += 1
                if args.trace:
                    assert len(rec) == len(ref_rec)
                    for i, ((name, shape, a), (_, _, b)) in enumerate(zip(rec, ref_rec)):
                        if not torch.equal(a, b):
                            firsts[name] = firsts.get(name, 0) + 1
                            print(f"   it {it}: first differing op #{i} " + _describe(name, shape, a, b), flush=True)
                            print("        before: " + " ".join(f"{r[0]}{r[1]}" for r in rec[max(i - 4, 0):i]), flush=True)
                            break
                elif bad <= 3:
                    du = int((cur[1] != ref[1]).sum()); df = int((cur[2] != ref[2]).sum())
                    print(f"   it {it}: loss {cur[0]!r} vs {ref[0]!r}, {du} unet / {df} fstext gradient words differ", flush=True)
                    for P, c, r, tag in ((tr.pu, cur[1], ref[1], "unet"), (tr.pf, cur[2], ref[2], "fstext")):
                        rows = []
                        for k in P.names:
                            a, b = P.view(c, k), P.view(r, k)
                            n = int((a != b).sum())
                            if n:
                                rel = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
                                rows.append(f"{k}:{n}/{a.numel()}:{rel:.1e}")
                        same = [k for k in P.names if not int((P.view(c, k) != P.view(r, k)).sum())]
                        print(f"     {tag}: {len(rows)} tensors differ, {len(same)} equal; equal: {same[:40]}", flush=True)
                        print("     differ: " + "  ".join(rows[:400]), flush=True)
        print(f"[train] cotenant={args.cotenant} graph={use_graph}: {bad} of {args.iters - 1} repeats differ from the first "
              f"({time.time() - t0:.1f} s)", flush=True)
    for k, v in sorted(firsts.items(), key=lambda kv: -kv[1]):
        print(f"[train]   first differing op {k}: {v} times", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--cotenant", type=int, default=1)
    ap.add_argument("--graph-first", type=int, default=0)
    ap.add_argument("--trace", type=int, default=0)
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
    rc = subprocess.call([sys.executable, __file__, "--role", "main", "--iters", str(args.iters), "--cotenant", str(args.cotenant), "--graph-first", str(args.graph_first), "--trace", str(args.trace)])
    stop.write_text("stop")
    if noise is not None:
        try:
            noise.wait(timeout=120)
        except subprocess.TimeoutExpired:
            noise.kill()
    sys.exit(rc)


if __name__ == "__main__":
    main()
