#!/bin/bash
# round 5, call E: the whole GPU suite with durations (target: well under 600 s) + smoke
mkdir -p gpurun_out/r05e
python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r05e/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05e/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05e/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r05e/smoke.log
tail -40 gpurun_out/r05e/gpu_tests.log; tail -5 gpurun_out/r05e/smoke.log
