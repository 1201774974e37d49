#!/bin/bash
# tile x split-K sweep of the step's conv shapes through the C ABI (build/lab_gemm): one line per (shape, tile, splits)
# usage: scripts/sweep_lab_convs.sh [LAB_ONLY pattern=conv] > log
PAT=${1:-conv}
for tile in 0 5 6 7 8 10 12 14 16 17 18; do
  for sp in 0 1 2 3 4 6 8; do
    if [ $tile == 0 ] && [ $sp != 0 ]; then continue; fi
    LAB_ONLY="$PAT" LAB_SPLITS=$sp build/lab_gemm 20 $tile 2>&1 | grep -v "^seer\|^shape\|TOTAL" | sed "s/^/tile $tile splits $sp | /"
  done
done
