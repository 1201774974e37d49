#!/bin/bash
# per-tensor choice of the GroupNorm statistics form: accumulated up to SEER_FX_MAX_ROWS_PB rows per batch element
for w in sthv2 sthv2_512 bridge; do
  for pb in 4096 1000000 4096 1000000; do
    if [ $w == sthv2 ]; then A="--steps 30"; else A="--workload $w --steps 10 --warmup 3"; fi
    SEER_FX_MAX_ROWS_PB=$pb python bench.py $A --no-cpu-baseline --no-train 2>/dev/null | tail -1 | W=$w PB=$pb python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['W'], 'max_rows_pb', os.environ['PB'], d['ms_per_step'])"
  done
done
