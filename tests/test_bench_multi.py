"""Rehearsal of `bench.py --gpus N` (N = 2, 4) on ONE MI355X: the ranks share cuda:0 over gloo (RCCL refuses two ranks per device), the
launch line is the driver's (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`).  Checks the
N > 1 control flow -- independent samples first, then the sharded step as the headline -- and the contract of the JSON line;
the numbers mean nothing (two processes on one device)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,partition", [(2, "batch_groups2xframe_shards1"),      # CFG halves: no data-path collective
                                              (4, "batch_groups2xframe_shards2")])     # + frame shards: GN all-reduce, K|V all-gather
def test_bench_multi_rank_contract(world, partition):
    env = dict(os.environ, SEER_BENCH_SAME_DEVICE="1", SEER_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-train"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "strong" and j["unit"] == "steps/s" and j["value"] > 0
    assert abs(j["value"] - 1e3 / j["ms_per_step"]) < 1e-2 * j["value"]
    assert j["config"]["parallelism"].startswith(partition)
    ws = j["weak_scaling"]
    assert ws["scaling"] == "weak" and ws["value"] > 0 and abs(ws["value"] - world * 1e3 / ws["ms_per_step"]) < 1e-2 * ws["value"]
    assert j["vs_baseline"] is None and j["dtype"] == "bf16" and j["data"] == "synthetic"



def test_bench_spawns_its_own_ranks_without_a_launcher():
    """plain `python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset): bench.py starts the two ranks itself and the line says
    n_gpus == 2 -- it never files a one-GPU measurement under N"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SEER_BENCH_SAME_DEVICE="1", SEER_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-train"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["parallelism"].startswith("batch_groups2xframe_shards1")
