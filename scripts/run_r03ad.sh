#!/bin/bash
# A/B: the library compiled for gfx950 (xnack "any") vs gfx950:xnack-
mkdir -p gpurun_out/r03ad
O=gpurun_out/r03ad
R=$PWD
for rep in 1 2; do
for v in base xnackoff; do
  SEER_HIP_LIB=$R/build/variants/$v/libseer_hip.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null > $O/bench_${v}_$rep.json
  python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v', $rep, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['spatial_attention_block']['us_per_launch'], d['roofline']['spatial_attention_block'].get('in_step_us'))
PY
done
done
for v in base xnackoff; do
  LD_LIBRARY_PATH=$R/build/variants/$v build/lab_gemm 20 > $O/lab_gemm_$v.log 2>&1; echo "$v $(tail -1 $O/lab_gemm_$v.log)"
  LAB_VARIANTS=23 LD_LIBRARY_PATH=$R/build/variants/$v build/lab_attn 30 2>&1 | grep -A2 -E "spatial L0|temporal L0|spatial 64" | grep variant | cut -c1-150
done
