#!/bin/bash
# tile sweep of the step's plain GEMM shapes through the C ABI (build/lab_gemm): one line per (shape, tile)
# usage: scripts/sweep_lab_plain.sh > log
for tile in 0 5 6 7 8 10 11 12 13 14 15 16 17 18 0; do
  for pat in "proj" "qkv" "shortcut" "ff"; do
    LAB_ONLY="$pat" build/lab_gemm 20 $tile 2>&1 | grep -v "^seer\|^shape\|TOTAL" | sed "s/^/tile $tile | /"
  done
done
