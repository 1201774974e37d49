#!/bin/bash
# batched W^T refresh + fused GroupNorm-backward group stage: kernel tests, trainer tests, the step time
mkdir -p gpurun_out/r03q
O=gpurun_out/r03q
timeout 1200 python -m pytest tests/test_train_kernels.py tests/test_gpu_train.py tests/test_fstext.py tests/test_gpu_cotenant.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python scripts/bench_train.py 10 > $O/bench_train.json 2> $O/bench_train.err; cat $O/bench_train.json
python scripts/bench_train.py 10 > $O/bench_train2.json 2>> $O/bench_train.err; cat $O/bench_train2.json
