#!/bin/bash
mkdir -p gpurun_out/r03ac
LAB_VARIANTS=23 build/lab_attn 30 > gpurun_out/r03ac/lab_attn.log 2>&1
grep -E "^==|variant" gpurun_out/r03ac/lab_attn.log | cut -c1-200 | grep -A2 -E "spatial L0|temporal L0|spatial 64|cross L0|ragged Sq"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -2
