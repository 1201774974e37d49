"""Where the time of seer_ff_fused_c320 goes: the kernel rebuilt with parts of its chunk loop left out (FF_PROBE bit mask in
csrc/ff_fused.hip: wrong results, timing only), each variant as its own small shared object, timed at 24 576 rows.

    python scripts/lab_ff_probe.py [mask ...]
"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
M, Cc, inner = 24576, 320, 1280
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(s, generator=g) * sc).to(dev)
h, x = r(M, Cc).to(bf16), r(M, Cc).to(bf16)
y = torch.empty_like(x)
gamma, beta = 1 + 0.1 * r(Cc), 0.1 * r(Cc)
w1, b1 = r(2 * inner, Cc, sc=Cc ** -0.5).to(bf16), 0.1 * r(2 * inner)
wcat, bcat = r(Cc, Cc + inner, sc=(Cc + inner) ** -0.5).to(bf16), 0.1 * r(Cc)
from seervideoldm_amd import ops  # noqa: E402

w1f, wcf = ops.ff_fused_pack(w1, wcat)
flop = 2.0 * M * Cc * 2 * inner + 2.0 * M * (Cc + inner) * Cc
NAMES = {1: "W1 stream", 2: "[Wp|WpW2] stream", 4: "MFMAs", 8: "GELU", 16: "fragment reads"}
masks = [int(a) for a in sys.argv[1:]] or [0]
out = ROOT / "gpurun_out" / "ffprobe"
out.mkdir(parents=True, exist_ok=True)
for mask in masks:
    so = out / f"ff_probe_{mask}.so"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT / 'include'}",
                    f"-I{ROOT / 'seervideoldm_amd' / 'csrc'}", "-fno-gpu-rdc", f"-DFF_PROBE={mask}",
                    str(ROOT / "seervideoldm_amd" / "csrc" / "ff_fused.hip"), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    fn = lib.seer_ff_fused_c320
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    fn.argtypes = [vp, i32, vp, i32, vp, i32, i64, vp, vp, C.c_float, vp, vp, vp, vp, vp, i64, i32, vp, vp]
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = fn(h.data_ptr(), Cc, x.data_ptr(), Cc, y.data_ptr(), Cc, M, gamma.data_ptr(), beta.data_ptr(), 1e-5, w1f.data_ptr(),
                b1.data_ptr(), wcf.data_ptr(), bcat.data_ptr(), None, 0, 0, None, st)
        assert rc == 0, rc
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    left_out = ", ".join(n for b, n in NAMES.items() if mask & b) or "nothing (the kernel as shipped)"
    print(f"mask {mask:2d}  {us:8.2f} us  {flop / us * 1e-6:7.1f} TFLOP/s-equivalent   left out: {left_out}", flush=True)

# the timeline of the shipped kernel: stamps of the first and the last workgroup (-DFF_STAMPS build)
so = out / "ff_stamps.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT / 'include'}",
                f"-I{ROOT / 'seervideoldm_amd' / 'csrc'}", "-fno-gpu-rdc", "-DFF_STAMPS",
                str(ROOT / "seervideoldm_amd" / "csrc" / "ff_fused.hip"), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
fn = lib.seer_ff_fused_c320
fn.argtypes = [vp, i32, vp, i32, vp, i32, i64, vp, vp, C.c_float, vp, vp, vp, vp, vp, i64, i32, vp, vp]
for _ in range(3):
    assert fn(h.data_ptr(), Cc, x.data_ptr(), Cc, y.data_ptr(), Cc, M, gamma.data_ptr(), beta.data_ptr(), 1e-5, w1f.data_ptr(),
              b1.data_ptr(), wcf.data_ptr(), bcat.data_ptr(), None, 0, 0, None, st) == 0
    torch.cuda.synchronize()
buf = (C.c_longlong * (2 * 4 * 256))()
assert lib.seer_lab_ff_stamps(buf) == 0
import numpy as np
t = np.array(buf[:], dtype=np.int64).reshape(2, 4, 256)
for blk in (0, 1):
    for wv in (0, 3):
        s = (t[blk, wv] - t[blk, wv, 0]) * 10          # ns
        print(f"workgroup {'first' if blk == 0 else 'last'} wave {wv}: tile + constants in {s[1]} | phase 0 done {s[2]} | barrier {s[3]} | "
              f"LayerNorm done {s[4]} | chunk 0 done {s[6]} | loop done {s[63]} | last product done {s[64]} | y in T {s[65]} | rows stored {s[66]}  (ns)")
        cyc = int(t[blk, wv, 201] - t[blk, wv, 200])
        print(f"    shader clock over start .. loop done: {cyc} cycles in {s[63]} ns = {cyc / max(int(s[63]), 1):.2f} GHz")
        ch = s[6:63].reshape(19, 3)
        nxt = np.append(ch[1:, 0], s[63])
        d = np.stack([ch[:, 1] - ch[:, 0], ch[:, 2] - ch[:, 1], nxt - ch[:, 2]], 1)
        print("    per iteration, ns (median over 19): H, two halves (first half's GEGLU beside the second) %d | barrier %d | "
              "Y step with the second half's GEGLU, g write, next requests %d | iteration %d" % (*np.median(d, 0), np.median(d.sum(1))))
