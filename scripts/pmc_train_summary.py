"""Per-launch counters of the training kernels from separate rocprofv3 --pmc passes of scripts/pmc_train_step.py.

    python scripts/pmc_train_summary.py FETCH.csv WRITE.csv UTIL.csv > profiles/r01_pmc_train.json

Units as in scripts/pmc_summary.py (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on
gfx950 for wide coalesced reads; bytes are "beyond L2" (HBM + MALL)."""
import csv
import json
import sys
from collections import defaultdict

FAMS = ("seer_gemm_tn_kernel", "seer_attn_bwd_kernel", "seer_attn_kernel", "seer_gemm_kernel", "adamw_kernel", "colpartial_kernel",
        "colfinal_kernel", "gn_bwd_apply_kernel", "ln_bwd_rows_kernel", "transpose_vec_kernel", "tn_reduce_kernel",
        "geglu_bwd_kernel", "sumsq_kernel")


def per_kernel(path, counters):
    tot = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        c = r["Counter_Name"]
        if c not in counters:
            continue
        fam = next((f for f in FAMS if f in r["Kernel_Name"]), None)
        if fam:
            tot[fam][c][0] += 1
            tot[fam][c][1] += float(r["Counter_Value"])
    return tot


def main():
    fetch = per_kernel(sys.argv[1], ("FETCH_SIZE",))
    write = per_kernel(sys.argv[2], ("WRITE_SIZE",))
    util = per_kernel(sys.argv[3], ("MfmaUtil", "VALUBusy", "LdsUtil")) if len(sys.argv) > 3 else {}
    out = {"source": "rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE | MfmaUtil VALUBusy LdsUtil) over "
                     "scripts/pmc_train_step.py (2 eager fine-tuning steps at config 5)",
           "correction": "bytes beyond L2 = (2 * FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
    for fam in FAMS:
        if fam not in fetch:
            continue
        n = fetch[fam]["FETCH_SIZE"][0]
        f = fetch[fam]["FETCH_SIZE"][1] * 1024 * 2
        w = write.get(fam, {}).get("WRITE_SIZE", [0, 0.0])[1] * 1024
        k = {"launches": n, "fetch_bytes_per_launch": round(f / n), "write_bytes_per_launch": round(w / n),
             "bytes_beyond_l2_per_launch": round((f + w) / n)}
        for c in ("MfmaUtil", "VALUBusy", "LdsUtil"):
            if fam in util and c in util[fam]:
                k[c + "_avg_percent"] = round(util[fam][c][1] / max(1, util[fam][c][0]), 1)
        out["kernels"][fam] = k
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
