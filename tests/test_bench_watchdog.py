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


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` without a launcher spawns its N ranks itself (tests/test_bench_multi.py, GPU); with fewer devices
    than N -- here: none -- it exits non-zero before touching a device and prints NO line: a one-GPU number is never filed under N"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SEER_BENCH_SAME_DEVICE")}
    env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(root))
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "--gpus 8" in r.stderr
    # ... and under a launcher whose world size disagrees with --gpus it refuses as well
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=str(root))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
