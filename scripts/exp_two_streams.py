"""Experiment: do the two CFG halves of a denoising step overlap usefully when replayed as two hipGraphs on two streams?

Per-kernel start/drain overhead is ~5 us x ~700 kernels per step; the halves (uncond / cond) are independent end to end
(GroupNorm statistics are per batch element), so a second stream can fill the other's ramps and tails.

    python scripts/exp_two_streams.py [steps]
"""
import copy
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from seervideoldm_amd import SeerUNet, synth  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = dict(synth.SD15_UNET_CFG)
model = SeerUNet(**cfg).to(dev)
model.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(cfg), device=dev), strict=True)
model.prepare()
e1 = model._engine
x_T, x0_emb, c, uc = bench.build_inputs(dev)
x = torch.cat([x0_emb, x_T], 2)
x2 = torch.cat([x, x], 0).float().contiguous()
ctx2 = torch.cat([uc, c], 0).contiguous()
t2 = torch.full((2,), 981, device=dev, dtype=torch.long)
cf = x0_emb.shape[2]


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = e1.run(x2, t2, ctx2, cf, use_graph=True).clone()
ms_one = timeit(lambda: e1.run(x2, t2, ctx2, cf, use_graph=True), steps)
print(f"one graph, B=2:            {ms_one:7.3f} ms/step")

# second engine: same packed weights, private caches / graphs
e2 = copy.copy(e1)
e2._rot_cache, e2._kv_cache, e2._kv_key, e2._graphs, e2._rec = {}, {}, None, {}, None
e1b = copy.copy(e1)
e1b._rot_cache, e1b._kv_cache, e1b._kv_key, e1b._graphs, e1b._rec = {}, {}, None, {}, None
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb = x2[0:1].contiguous(), x2[1:2].contiguous()
ca, cb = ctx2[0:1].contiguous(), ctx2[1:2].contiguous()
ta, tb = t2[0:1].contiguous(), t2[1:2].contiguous()
outs = [None, None]


def two():
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur)
    sB.wait_stream(cur)
    with torch.cuda.stream(sA):
        outs[0] = e1b.run(xa, ta, ca, cf, use_graph=True)
    with torch.cuda.stream(sB):
        outs[1] = e2.run(xb, tb, cb, cf, use_graph=True)
    cur.wait_stream(sA)
    cur.wait_stream(sB)


two()
torch.cuda.synchronize()
got = torch.cat(outs, 0)
print("two-stream vs one-graph max |diff|:", float((got - ref).abs().max()), " bit-equal:", bool(torch.equal(got, ref)))
ms_two = timeit(two, steps)
print(f"two graphs (B=1) on 2 streams: {ms_two:7.3f} ms/step")


def seq():
    outs[0] = e1b.run(xa, ta, ca, cf, use_graph=True)
    outs[1] = e2.run(xb, tb, cb, cf, use_graph=True)


seq()   # graphs were captured on sA / sB; replaying them on the current stream is fine
ms_seq = timeit(seq, steps)
print(f"two graphs (B=1) back to back: {ms_seq:7.3f} ms/step")
