#!/bin/bash
mkdir -p gpurun_out/r03ae
O=gpurun_out/r03ae
R=$PWD
for rep in 1 2; do
for v in base noslp; do
  SEER_HIP_LIB=$R/build/variants/$v/libseer_hip.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null > $O/bench_${v}_$rep.json
  python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v', $rep, d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done
done
for v in base noslp; do
  LD_LIBRARY_PATH=$R/build/variants/$v build/lab_gemm 20 > $O/lab_gemm_$v.log 2>&1; echo "$v $(tail -1 $O/lab_gemm_$v.log)"
done
