"""Per-launch HBM traffic of the GEMM template from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/pmc_step.py.

    python scripts/pmc_summary.py gpurun_out/pmc_step_fetch/*/*_counter_collection.csv \
                                  gpurun_out/pmc_step_write/*/*_counter_collection.csv  > profiles/r02_pmc_traffic.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read stream on gfx950 -> doubled; WRITE_SIZE is
exact for 16-B-per-lane stores (our epilogue stores are 8 B per lane: uncalibrated, taken as is).
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        fam = "seer_rowchain_c320_kernel" if "seer_rowchain_c320_kernel" in name else "seer_ff_fused_c320_kernel" if "seer_ff_fused_c320_kernel" in name else "seer_gemm_ws_kernel" if "seer_gemm_ws_kernel" in name else "seer_gemm_t320_kernel" if "seer_gemm_t320_kernel" in name else "seer_gemm_kernel" if "seer_gemm_kernel" in name else (
            "seer_attn40_kernel" if "seer_attn40_kernel" in name else ("seer_attn_kernel" if "seer_attn_kernel" in name else None))
        if fam is None:
            for k in ("gn_stats", "gn_apply", "gn_finalize", "layernorm", "splitk_reduce"):
                if k in name:
                    fam = k
        if fam:
            tot[fam][0] += 1
            tot[fam][1] += float(r["Counter_Value"])
    return tot


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    from pathlib import Path
    sha = (Path(__file__).resolve().parents[1] / "seervideoldm_amd" / "lib" / "build.sha256").read_text().strip()
    out = {"build_sha256": sha,      # bench.py refuses the file when the library has been rebuilt since
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/pmc_step.py "
                     "(2 eager full-size denoising steps, config 2)",
           "correction": "HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           "kernels": {}}
    for fam in fetch:
        n, f = fetch[fam]
        nw, w = write.get(fam, [0, 0.0])
        out["kernels"][fam] = {"launches": n, "fetch_kib_per_launch": f / n, "write_kib_per_launch": (w / nw if nw else None),
                               "hbm_bytes_per_launch": (2 * f / n + (w / nw if nw else 0.0)) * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
