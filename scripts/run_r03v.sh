#!/bin/bash
# tile A/B with the real epilogues (lab_gemm) on the rows the B=2 F=12 sweep flagged
mkdir -p gpurun_out/r03v
O=gpurun_out/r03v/lab_tiles.log
: > $O
for rep in 1 2; do
for t in 0 12 16 5 7; do
  echo "== tile override $t (rep $rep)" >> $O
  LAB_ONLY="qkv" build/lab_gemm 30 $t 2>&1 | grep -E "^qkv" >> $O
  LAB_ONLY="ff1 geglu" build/lab_gemm 30 $t 2>&1 | grep -E "^ff1" >> $O
  LAB_ONLY="shortcut L0" build/lab_gemm 30 $t 2>&1 | grep -E "^shortcut" >> $O
done
done
for s in 0 8 16; do echo "== conv 4x4 2560 splits $s" >> $O; LAB_SPLITS=$s LAB_ONLY="conv 4x4 2560" build/lab_gemm 30 2>&1 | grep -E "^conv" >> $O; done
cat $O
