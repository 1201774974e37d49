#!/bin/bash
# round 4: LDS-direct ring tiles with hoisted staging offsets (conv gather by tap bitmask, plain operands by buffer loads): parity
# tests of every GEMM / conv kernel test, then the step table
mkdir -p gpurun_out/r04
L=gpurun_out/r04/convfast.log
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_t320.py tests/test_gpu_f16.py -k "gemm or conv or colsums or geglu or rotary" -x -q 2>&1 | tail -3 > $L
timeout 300 build/lab_gemm 20 0 >> $L 2>&1
head -3 $L; grep -E "^ff|^qkv|^proj|^shortcut|TOTAL" $L
