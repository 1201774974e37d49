#!/bin/bash
mkdir -p gpurun_out/r03y
LAB_VARIANTS=23 build/lab_attn 30 > gpurun_out/r03y/lab_attn_qb2.log 2>&1
grep -E "^==|variant" gpurun_out/r03y/lab_attn_qb2.log | cut -c1-200 | head -16
grep -A2 "spatial 64" gpurun_out/r03y/lab_attn_qb2.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -2
