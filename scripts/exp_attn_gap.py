"""The d = 40 spatial attention block launched back to back (one replayed HIP graph of 25 launches, then 25 eager launches):
run under `rocprofv3 --kernel-trace --output-format csv` and feed the kernel trace to this script's `--summarise` mode to get
the kernel durations beside the start-to-start intervals (what a launch costs including the boundary behind it).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -- python3 $R/scripts/exp_attn_gap.py
    (ATTN_DATA=randn|small|const|zeros picks the q|k|v contents: the kernel's duration follows the operand bit activity)
    python scripts/exp_attn_gap.py --summarise out/*/*_kernel_trace.csv
"""
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def summarise(path):
    rows = [r for r in csv.DictReader(open(path)) if "attn40" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
    gap = [g for g in gap if g < 200]          # drop the pauses between the phases of the script
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{len(rows)} launches: kernel duration median {med(dur):.1f} us (min {min(dur):.1f}, max {max(dur):.1f}); "
          f"end-to-next-start gap median {med(gap):.1f} us (min {min(gap):.1f}, max {max(gap):.1f})")
    wg = {r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "?") for r in rows}
    keys = [k for k in rows[0].keys() if "LDS" in k or "Scratch" in k or "Grid" in k or "Workgroup" in k or "VGPR" in k or "SGPR" in k]
    print({k: rows[0][k] for k in keys})


def main():
    import torch
    from seervideoldm_amd import ops
    dev = torch.device("cuda:0")
    B, S, H, d = 24, 1024, 8, 40
    C = H * d
    import os
    kind = os.environ.get("ATTN_DATA", "randn")           # randn | small (randn / 16) | const (all 0.5) | zeros
    qkv = torch.randn((B * S, 3 * C), device=dev)
    qkv = {"randn": qkv, "small": qkv / 16, "const": torch.full_like(qkv, 0.5), "zeros": torch.zeros_like(qkv)}[kind].to(torch.bfloat16)
    out = torch.empty((B * S, C), device=dev, dtype=torch.bfloat16)
    run = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, batch=B, heads=H, head_dim=d, Sq=S, Sk=S)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(25):
            run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        main()
