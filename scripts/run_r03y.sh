#!/bin/bash
mkdir -p gpurun_out/r03y
LAB_VARIANTS=23 build/lab_attn 30 > gpurun_out/r03y/lab_attn_qb2.log 2>&1
grep -E "^==|variant" gpurun_out/r03y/lab_attn_qb2.log | cut -c1-200
timeout 600 python -m pytest tests/test_bench_line.py -m gpu -x -q 2>&1 | tail -2
