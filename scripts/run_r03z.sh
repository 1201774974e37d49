#!/bin/bash
mkdir -p gpurun_out/r03z
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q -k "attention or alternate or full_size or 512" > gpurun_out/r03z/pytest.log 2>&1; tail -3 gpurun_out/r03z/pytest.log
for w in sthv2_512 bridge; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | cut -c1-260; done
SEER_X=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | cut -c1-200
