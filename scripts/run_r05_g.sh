#!/bin/bash
# round 5, call G: ff.net.2 and proj_out as one two-source GEMM -- parity, then A/B of the step on one box
mkdir -p gpurun_out/r05g
python -m pytest tests/test_gpu_unet.py tests/test_gpu_golden_w320.py tests/test_dist_gpu.py -q -x > gpurun_out/r05g/tests.log 2>&1; echo "rc $?" >> gpurun_out/r05g/tests.log
for i in 1 2; do
SEER_FF_FOLD=0 python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/r05g/bench_nofold_$i.json.log 2>/dev/null
SEER_FF_FOLD=1 python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/r05g/bench_fold_$i.json.log 2>/dev/null
done
tail -4 gpurun_out/r05g/tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05g/bench_*.json.log')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], r['frac'], r['launches_per_step'], r['step_breakdown_ms']['gemm'])
PY
