"""Per-kernel-family averages of the derived utilisation counters of one rocprofv3 --pmc pass over scripts/pmc_step.py
(MfmaUtil VALUBusy LdsUtil OccupancyPercent), weighted by nothing: one sample per launch.

    python scripts/pmc_util_summary.py <..._counter_collection.csv>  > profiles/r02_pmc_utilisation.json
"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(seer_gemm_kernel)<(\d+), (\d+), (true|false), (true|false), (true|false), (\d+)", name)
    if m:
        return f"gemm<{m.group(2)}x{m.group(3)}{', conv' if m.group(4) == 'true' else ''}{', geglu' if m.group(5) == 'true' else ''}" \
               f"{', split-K' if m.group(6) == 'true' else ''}, {m.group(7)} stages>"
    return re.sub(r"[<(].*", "", name).strip()


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(sys.argv[1])):
        f = family(r["Kernel_Name"])
        if f.startswith("at::") or "rocclr" in f or "elementwise" in f:
            continue
        a = acc[f][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    out = {}
    for f, cs in sorted(acc.items()):
        out[f] = {"launches": max(v[0] for v in cs.values())}
        for c, (n, tot) in cs.items():
            out[f][c] = round(tot / n, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
