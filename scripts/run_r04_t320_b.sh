#!/bin/bash
# round 4: 256 x 320 tile kernel after the conv-addressing / residual-staging fixes; config-2 and config-4 (4x rows) shape tables
mkdir -p gpurun_out/r04
L=gpurun_out/r04/t320_b.log
echo "== check (tile 22 vs AUTO), column sums on" > $L
LAB_CHECK=1 LAB_COLSUM=1 timeout 600 build/lab_gemm 3 22 2>&1 | grep -B1 -E "MISMATCH|rc [1-9-]" >> $L
echo "== config 2, tile 22" >> $L
timeout 300 build/lab_gemm 20 22 >> $L 2>&1
echo "== config 4 rows (x4), tile 22" >> $L
LAB_MMUL=4 timeout 600 build/lab_gemm 10 22 >> $L 2>&1
echo "== config 4 rows (x4), AUTO" >> $L
LAB_MMUL=4 timeout 600 build/lab_gemm 10 0 >> $L 2>&1
echo "== config 4 rows (x4), tile 22, no split" >> $L
LAB_MMUL=4 LAB_SPLITS=1 timeout 600 build/lab_gemm 10 22 >> $L 2>&1
tail -3 $L
