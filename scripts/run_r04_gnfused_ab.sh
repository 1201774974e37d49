#!/bin/bash
# round 4: A/B of the fused GroupNorm (statistics from column sums inside the apply launch, 16x16 level down) in the default bench
mkdir -p gpurun_out/r04
L=gpurun_out/r04/gnfused_ab.log
: > $L
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "groupnorm" -x -q 2>&1 | tail -2 >> $L
timeout 1800 python -m pytest tests/test_gpu_unet.py -x -q 2>&1 | tail -3 >> $L
for v in 1 0 1 0; do
  SEER_GN_FUSED=$v timeout 600 python bench.py --no-cpu-baseline --no-train 2>/dev/null | python -c "
import json, sys
l=[x for x in sys.stdin if x.startswith('{')][-1]
d=json.loads(l); b=d['roofline']['step_breakdown_ms']; print('SEER_GN_FUSED=$v', d['value'], d['ms_per_step'], 'stats', b.get('groupnorm_stats'), 'apply', b.get('groupnorm_apply'))" >> $L
done
python scripts/exp_gn_fused.py 2>&1 | tail -15 >> $L
cat $L
