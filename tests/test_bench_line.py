"""The contract of `python bench.py` at N = 1 (the command the driver runs): ONE JSON line with the metric, the whole-job value, the
`roofline` object of the dominant kernel class and the `cpu_baseline` object -- checked on a short run (few steps, small CPU budget;
the numbers are not asserted, their consistency is)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_single_gpu_bench_line_contract():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--no-train", "--cpu-budget-s", "3"],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "steps/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                                   # BASELINE.md publishes no number for this metric
    assert abs(d["value"] * d["ms_per_step"] - 1e3) < 1.0             # one sample: steps/s = 1000 / ms per step
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["algorithmic_gflop_per_launch"] / rf["avg_launch_us"] * 1e3) < 0.02 * rf["achieved"]
    # traffic: the PMC figure of THIS build (profiles/r03_pmc_traffic.json carries the library digest) or null with the reason
    assert ("traffic" in rf) and (rf["traffic"] is None and "traffic_note" in rf or rf["traffic"] > rf["algorithmic_bytes_per_launch"] * 0.5)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "steps/s" and cb["cores"] >= 1 and 0 < cb["value"] < d["value"] and cb["sample"]
