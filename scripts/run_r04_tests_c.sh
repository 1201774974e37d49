#!/bin/bash
# round 4: fused GroupNorm (statistics from column sums inside the apply launch): kernel test, UNet parity, bench A/B
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "groupnorm" -x -q 2>&1 | tail -5 > gpurun_out/r04/tests_gnfused.log
timeout 1800 python -m pytest tests/test_gpu_unet.py -x -q 2>&1 | tail -5 >> gpurun_out/r04/tests_gnfused.log
timeout 600 python bench.py --no-cpu-baseline --no-train > gpurun_out/r04/bench_gnfused.json.log 2>gpurun_out/r04/bench_gnfused.err
cat gpurun_out/r04/tests_gnfused.log; python -c "
import json
l=[x for x in open('gpurun_out/r04/bench_gnfused.json.log') if x.startswith('{')][-1]
d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['step_breakdown_ms'])"
