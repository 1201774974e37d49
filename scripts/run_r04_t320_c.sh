#!/bin/bash
# round 4: the 256 x 320 tile after the staging path became scalar (buffer-resource LDS-DMA, per-lane offsets hoisted): check, then the
# config-4 and config-2 tables for tile 22 and AUTO
mkdir -p gpurun_out/r04
L=gpurun_out/r04/t320_c.log
echo "== check (tile 22 vs AUTO tile), column sums on" > $L
LAB_CHECK=1 LAB_COLSUM=1 timeout 600 build/lab_gemm 3 22 2>&1 | grep -B1 -E "MISMATCH|rc [1-9-]" >> $L
echo "== config 4 rows (x4), tile 22, no split" >> $L
LAB_MMUL=4 LAB_SPLITS=1 timeout 600 build/lab_gemm 10 22 >> $L 2>&1
echo "== config 4 rows (x4), AUTO" >> $L
LAB_MMUL=4 timeout 600 build/lab_gemm 10 0 >> $L 2>&1
echo "== config 2, tile 22, no split" >> $L
LAB_SPLITS=1 timeout 300 build/lab_gemm 20 22 >> $L 2>&1
echo "== config 2, AUTO" >> $L
timeout 300 build/lab_gemm 20 0 >> $L 2>&1
grep -E "^==|TOTAL|MISMATCH" $L
