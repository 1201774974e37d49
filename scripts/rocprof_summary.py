"""Aggregate a rocprofv3 `--kernel-trace --stats` kernel_stats.csv by kernel family (template arguments stripped).

    python scripts/rocprof_summary.py gpurun_out/rocprof_xx/<host>/<pid>_kernel_stats.csv [steps] > profiles/xx_summary.md

`steps` = number of denoising steps the profiled command ran (warm-up included) to print per-step figures.
The `seer_gemm_kernel` row is the one bench.py's `roofline.avg_launch_us` must agree with.
"""
import csv
import re
import sys
from collections import defaultdict


def family(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"<.*", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.strip()


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fam = defaultdict(lambda: [0, 0])
    inst = []
    for row in csv.DictReader(open(path)):
        calls, tot = int(row["Calls"]), int(row["TotalDurationNs"])
        f = family(row["Name"])
        fam[f][0] += calls
        fam[f][1] += tot
        inst.append((row["Name"], calls, tot))
    total = sum(v[1] for v in fam.values())
    print(f"# rocprofv3 kernel summary ({path})\n")
    print("| kernel family | calls | total ms | avg us | % of GPU time |" + (" ms/step |" if steps else ""))
    print("|---|---:|---:|---:|---:|" + ("---:|" if steps else ""))
    for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        if t / total < 0.002:
            continue
        line = f"| `{f}` | {c} | {t / 1e6:.3f} | {t / c / 1e3:.2f} | {100 * t / total:.1f} |"
        if steps:
            line += f" {t / 1e6 / steps:.3f} |"
        print(line)
    print("\nTop instantiations of `seer_gemm_kernel<BM, BN, CONV, GEGLU, SPLIT, STAGES>`:\n")
    print("| instantiation | calls | avg us | total ms |")
    print("|---|---:|---:|---:|")
    for n, c, t in sorted(inst, key=lambda x: -x[2]):
        if "seer_gemm_kernel" in n or "seer_gemm_t320_kernel" in n:
            m = re.search(r"seer_gemm(_t320)?_kernel<([^>]*)>", n)
            print(f"| `{'t320' if m and m.group(1) else ''}<{m.group(2) if m else '?'}>` | {c} | {t / c / 1e3:.2f} | {t / 1e6:.3f} |")


if __name__ == "__main__":
    main()
