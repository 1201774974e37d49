#!/bin/bash
# A/B the attention kernel variants on ONE device (timings across gpurun boxes differ by up to 12 %).
for dbuf in 0 1; do for thr in 0 4; do
  echo "== SEER_ATTN_DBUF=$dbuf SEER_ATTN_DEFER=$thr"
  SEER_ATTN_DBUF=$dbuf SEER_ATTN_DEFER=$thr python scripts/bench_kernels.py attn | grep attn
done; done
