#!/bin/bash
# round 4: first runs of the 256 x 320 tile kernel (tile code 22) through the C ABI harness
mkdir -p gpurun_out/r04
echo "== check (tile 22 vs AUTO), column sums on" > gpurun_out/r04/t320_a.log
LAB_CHECK=1 LAB_COLSUM=1 timeout 600 build/lab_gemm 5 22 >> gpurun_out/r04/t320_a.log 2>&1
echo "== timing tile 22" >> gpurun_out/r04/t320_a.log
timeout 300 build/lab_gemm 20 22 >> gpurun_out/r04/t320_a.log 2>&1
echo "== timing AUTO" >> gpurun_out/r04/t320_a.log
timeout 300 build/lab_gemm 20 0 >> gpurun_out/r04/t320_a.log 2>&1
tail -5 gpurun_out/r04/t320_a.log
