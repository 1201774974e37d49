#!/bin/bash
# LayerNorm folded into the consuming GEMM (seer_gemm_desc::rowstat / ln_rowstat): tests, then the step with / without (SEER_LN_FOLD)
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "row_statistics or folded" 2>&1 | tail -12
python -m pytest tests/test_gpu_unet.py -q -m gpu -x 2>&1 | tail -5
for v in 1 0 1 0; do
  SEER_LN_FOLD=$v python bench.py --no-cpu-baseline --no-train --steps 30 2>/dev/null | V=$v python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEER_LN_FOLD=' + os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['step_breakdown_ms'])"
done
