#!/bin/bash
# round 4: recalibration of the AUTO rule after the small tiles' staging path got faster: config-4 rows, small tiles only (tile 20) vs
# the 256 x 320 tile unsplit (tile 22, splits 1) vs AUTO
mkdir -p gpurun_out/r04
L=gpurun_out/r04/t320_d.log
echo "== config 4 rows (x4), small tiles only (tile 20)" > $L
LAB_MMUL=4 timeout 600 build/lab_gemm 10 20 >> $L 2>&1
echo "== config 4 rows (x4), tile 22, no split" >> $L
LAB_MMUL=4 LAB_SPLITS=1 timeout 600 build/lab_gemm 10 22 >> $L 2>&1
echo "== config 4 rows (x4), AUTO" >> $L
LAB_MMUL=4 timeout 600 build/lab_gemm 10 0 >> $L 2>&1
echo "== config 2, small tiles only (tile 20)" >> $L
timeout 300 build/lab_gemm 20 20 >> $L 2>&1
echo "== config 2, tile 22, no split" >> $L
LAB_SPLITS=1 timeout 300 build/lab_gemm 20 22 >> $L 2>&1
grep -E "^==|TOTAL" $L
