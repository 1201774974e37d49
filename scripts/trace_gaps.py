"""Where the time of a replayed denoising step goes, from a rocprofv3 --kernel-trace CSV of `bench.py`:

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-train
    python scripts/trace_gaps.py out/*/*_kernel_trace.csv

A step = the kernels from one `ddim_assemble_kernel` (first node of the captured step) to the next.  For the steady-state steps
it prints the step span, the sum of kernel durations, the idle time between kernels (gap = next start - previous end), the
number of kernels, and the kernel families ranked by time and by the gap in front of them."""
import csv
import re
import sys
from collections import defaultdict


def fam(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(seer_gemm_kernel)<(\d+), (\d+), (true|false), (true|false), (true|false), (\d+)", name)
    if m:
        return f"gemm<{m.group(2)}x{m.group(3)}{',conv' if m.group(4) == 'true' else ''}{',geglu' if m.group(5) == 'true' else ''}" \
               f"{',splitK' if m.group(6) == 'true' else ''},{m.group(7)}>"
    m = re.match(r"(seer_attn40_kernel)<(\d+), (true|false), (true|false)", name)
    if m:
        return "attn40<" + ("plain" if m.group(4) == "true" else "window/causal") + ">"
    return re.sub(r"[<(].*", "", name).strip()


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "ddim_assemble_kernel" in r[2]]
    if len(starts) < 4:
        sys.exit("fewer than 4 captured steps in the trace")
    steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
    # steady state: the most common kernel count per step
    counts = defaultdict(int)
    for a, b in steps:
        counts[b - a] += 1
    n_common = max(counts, key=counts.get)
    steady = [(a, b) for a, b in steps if b - a == n_common][2:]
    print(f"{len(steps)} steps in the trace, {len(steady)} steady-state steps of {n_common} kernels each")
    span = busy = gap = 0.0
    t_f, g_f, c_f = defaultdict(float), defaultdict(float), defaultdict(int)
    for a, b in steady:
        span += rows[b][0] - rows[a][0]
        prev_end = rows[a][0]
        for i in range(a, b):
            s, e, n = rows[i]
            f = fam(n)
            busy += e - s
            g = max(0, s - prev_end)
            gap += g
            t_f[f] += e - s
            g_f[f] += g
            c_f[f] += 1
            prev_end = max(prev_end, e)
        gap += max(0, rows[b][0] - prev_end)
    k = len(steady)
    print(f"per step: span {span / k / 1e6:.3f} ms, kernels busy {busy / k / 1e6:.3f} ms, idle between kernels {gap / k / 1e6:.3f} ms "
          f"({100 * gap / span:.1f} %)")
    print("\n| kernel family | per step | ms/step | avg us | idle in front, us avg |")
    print("|---|---:|---:|---:|---:|")
    for f in sorted(t_f, key=lambda x: -t_f[x]):
        print(f"| `{f}` | {c_f[f] / k:.0f} | {t_f[f] / k / 1e6:.3f} | {t_f[f] / c_f[f] / 1e3:.2f} | {g_f[f] / c_f[f] / 1e3:.2f} |")


if __name__ == "__main__":
    main()
