"""bench.py's safety net for N > 1 (CPU test: no GPU in bench.Watchdog): a phase with collectives that never returns must still leave
the driver one JSON line."""
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

def test_watchdog_prints_the_fallback_line_and_exits_nonzero():
    """a phase that never returns (a hung collective): at the deadline rank 0 prints the fallback JSON line and the process leaves
    with exit code 3 (the job did not finish; the line is still there for the driver); a disarmed watchdog does nothing.  CPU only: bench.Watchdog has no GPU in it."""
    import json
    import subprocess
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench.Watchdog(int(sys.argv[1])); d.arm(0.3, lambda: {'value': 7, 'scaling': 'weak'}); d.disarm()\n"
            "d.arm(0.5, lambda: {'value': 42, 'scaling': 'weak'})\n"
            "time.sleep(30); print('never')\n") % str(ROOT)
    for rank, want in ((0, {"value": 42, "scaling": "weak"}), (1, None)):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and time.time() - t0 < 25, (r.returncode, r.stderr[-500:])
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert "never" not in r.stdout
        assert (json.loads(lines[-1]) == want and len(lines) == 1) if want is not None else lines == []
