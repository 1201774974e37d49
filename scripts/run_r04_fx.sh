#!/bin/bash
# fixed-point accumulated column sums (seer_gemm_desc::colsum_fx): tests, then the step with / without (SEER_GN_FX)
mkdir -p gpurun_out/fx
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "fixed_point or apply_fx or colsum or column" 2>&1 | tail -8
python -m pytest tests/test_gpu_unet.py -q -m gpu -x 2>&1 | tail -5
for v in 1 0 1 0; do
  SEER_GN_FX=$v python bench.py --no-cpu-baseline --no-train --steps 30 2>/dev/null | V=$v python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEER_GN_FX=' + os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['step_breakdown_ms'])"
done
