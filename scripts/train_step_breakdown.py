"""One steady-state fine-tuning step out of a rocprofv3 --kernel-trace CSV of scripts/bench_train.py: kernels between two `sumsq_kernel`
launches (the optimizer's gradient norm: one per step), by family.

    python scripts/train_step_breakdown.py gpurun_out/r06t/rocprof/*/*_kernel_trace.csv > profiles/r06_train_step_profile.md
"""
import csv
import re
import sys
from collections import defaultdict


def fam(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(seer_gemm_kernel)<(\d+), (\d+), (true|false), (true|false), (true|false), (\d+)", n)
    if m:
        return (f"gemm<{m.group(2)}x{m.group(3)}{',conv' if m.group(4) == 'true' else ''}{',geglu' if m.group(5) == 'true' else ''}"
                f"{',splitK' if m.group(6) == 'true' else ''},{m.group(7)}>")
    m = re.match(r"seer_attn_bwd_kernel<(\d+), (\d)>", n)
    if m:
        return f"seer_attn_bwd_kernel<d{m.group(1)}, {'dK|dV' if m.group(2) == '1' else 'dQ'}>"
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+(\w+?_kernel).*", r"\1", n)
    return re.sub(r"[<(].*", "", n).strip()[:70]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    st = [i for i, r in enumerate(rows) if "sumsq_kernel" in r[2]]
    if len(st) < 4:
        sys.exit("fewer than 4 optimizer steps in the trace")
    a, b = st[-3], st[-2]
    t, c = defaultdict(float), defaultdict(int)
    busy = 0
    for i in range(a, b):
        s, e, n = rows[i]
        f = fam(n)
        t[f] += e - s
        c[f] += 1
        busy += e - s
    print(f"one steady-state step: {b - a} kernels, {busy / 1e6:.3f} ms of kernel time (the profiler's own serialisation stretches the span "
          f"to {(rows[b][0] - rows[a][0]) / 1e6:.1f} ms; bench_train.py without it is the step time)\n")
    print("| kernel family | launches | ms / step | avg us |\n|---|---:|---:|---:|")
    for f in sorted(t, key=lambda f: -t[f]):
        print(f"| `{f}` | {c[f]} | {t[f] / 1e6:.3f} | {t[f] / c[f] / 1e3:.2f} |")


main()
