#!/bin/bash
# round 5, call D: the exact-statistics kernel and the sharded step on HIP (2 processes on one GPU), the rank shapes of every N > 1
# configuration, the fixture-based full-size tests
mkdir -p gpurun_out/r05d
python -m pytest tests/test_gpu_kernels.py -q -x -k "stats_fx or groupnorm" > gpurun_out/r05d/gn_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05d/gn_tests.log
python -m pytest tests/test_dist_gpu.py tests/test_abi.py -q -x > gpurun_out/r05d/dist_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05d/dist_tests.log
python -m pytest tests/test_gpu_unet.py -q -x --durations=12 > gpurun_out/r05d/unet_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05d/unet_tests.log
python scripts/exp_shard_sizes.py > gpurun_out/r05d/shard_sizes.log 2>&1
tail -3 gpurun_out/r05d/gn_tests.log; tail -5 gpurun_out/r05d/dist_tests.log; tail -18 gpurun_out/r05d/unet_tests.log; cat gpurun_out/r05d/shard_sizes.log
