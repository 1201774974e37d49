#!/bin/bash
# round 5, call A: GPU suite with durations (which tests spend the time), the d = 40 attention lab table on the starting kernels,
# one short bench line
mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/r05a/gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05a/gpu_tests.log
LAB_VARIANTS=23 build/lab_attn 30 > gpurun_out/r05a/lab_attn.log 2>&1
python bench.py --steps 20 --warmup 3 --no-train --cpu-budget-s 5 > gpurun_out/r05a/bench.json.log 2> gpurun_out/r05a/bench.err
tail -5 gpurun_out/r05a/gpu_tests.log
