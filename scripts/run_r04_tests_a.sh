#!/bin/bash
# round 4: GPU tests (new 256x320 tile tests first), then the default bench line
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_t320.py -x -q 2>&1 | tail -15 > gpurun_out/r04/tests_t320.log
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04/tests_gpu_a.log
timeout 900 python bench.py > gpurun_out/r04/bench_a.json.log 2>gpurun_out/r04/bench_a.err
tail -3 gpurun_out/r04/tests_t320.log gpurun_out/r04/tests_gpu_a.log; tail -c 600 gpurun_out/r04/bench_a.json.log
