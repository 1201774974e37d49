#!/bin/bash
mkdir -p gpurun_out/r03t
O=gpurun_out/r03t
timeout 1500 python -m pytest tests/test_bench_multi.py tests/test_dist_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -2 $O/bench.err
