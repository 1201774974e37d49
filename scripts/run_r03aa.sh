#!/bin/bash
mkdir -p gpurun_out/r03aa
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > gpurun_out/r03aa/bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03aa/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['spatial_attention_block'], d['roofline']['frac'])
PY
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_golden_w320.py -m gpu -x -q 2>&1 | tail -2
