#!/bin/bash
# round 5, call C: the ring form of the d = 40 attention as the default + the reference-fixture tests at full size
mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" > gpurun_out/r05c/attn_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05c/attn_tests.log
python -m pytest tests/test_gpu_unet.py tests/test_gpu_lifecycle.py -q -x --durations=15 > gpurun_out/r05c/unet_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05c/unet_tests.log
python bench.py --steps 20 --warmup 3 --no-train --cpu-budget-s 3 > gpurun_out/r05c/bench.json.log 2> gpurun_out/r05c/bench.err
tail -3 gpurun_out/r05c/attn_tests.log; tail -22 gpurun_out/r05c/unet_tests.log
