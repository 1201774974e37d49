#!/bin/bash
mkdir -p gpurun_out/r03r
O=gpurun_out/r03r
python scripts/sweep_small.py > $O/sweep_small.log 2>&1; cat $O/sweep_small.log
timeout 600 python -m pytest tests/test_train_kernels.py -m gpu -x -q -k "groupnorm or gn" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python scripts/bench_train.py 10 > $O/bench_train.json 2> $O/bench_train.err; cut -c1-330 $O/bench_train.json
