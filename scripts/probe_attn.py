"""Which resource bounds the flash-attention kernel?  Times the step's attention shapes with the measurement-only library
variants built by scripts/probe_attn.sh (one resource removed at a time; results are garbage).

    python scripts/probe_attn.py
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
VARIANTS = [(0, "full"), (1, "no-exp"), (2, "no-PV-mfma"), (4, "no-QK-mfma"), (6, "no-mfma"), (8, "tile-staged-once"),
            (9, "no-exp+staged-once"), (15, "softmax-VALU-only")]
CASES = [("spatial L0 d40", 24, 1024, 1024, 40, False), ("spatial 64^2 d40", 24, 4096, 4096, 40, False),
         ("spatial L1 d80", 24, 256, 256, 80, False), ("flat causal d40", 64, 768, 768, 40, True)]


def child():
    import torch
    sys.path.insert(0, str(ROOT))
    from seervideoldm_amd import ops
    dev = torch.device("cuda:0")
    bf16 = torch.bfloat16
    out = []
    for name, B, Sq, Sk, d, causal in CASES:
        C = 8 * d
        q = torch.randn(B * Sq, C, device=dev).to(bf16)
        k = torch.randn(B * Sk, C, device=dev).to(bf16)
        v = torch.randn(B * Sk, C, device=dev).to(bf16)
        o = torch.empty(B * Sq, C, device=dev, dtype=bf16)
        fn = lambda: ops.attention(q, k, v, o, batch=B, heads=8, head_dim=d, Sq=Sq, Sk=Sk, causal=causal)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / 20 * 1e3)
    print(" ".join(f"{t:9.1f}" for t in out))


def main():
    print("variant".ljust(22) + " ".join(c[0].rjust(9)[-9:] for c in CASES) + "   (us)")
    for bits, name in VARIANTS:
        env = dict(os.environ)
        if bits:
            env["SEER_HIP_LIB"] = str(ROOT / "seervideoldm_amd" / "lib" / f"libseer_aprobe{bits}.so")
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.strip() and l.strip()[0].isdigit()]
        print(name.ljust(22) + (line[-1] if line else "FAILED " + r.stderr[-300:]))


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else main()
