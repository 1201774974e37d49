#!/bin/bash
# round 4: the conv gather of the LDS-direct ring tiles without per-piece address arithmetic: parity tests, then the step table
mkdir -p gpurun_out/r04
L=gpurun_out/r04/convfast.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "conv or colsums" -x -q 2>&1 | tail -3 > $L
timeout 300 build/lab_gemm 20 0 >> $L 2>&1
cat $L | tail -40
