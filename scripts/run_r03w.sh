#!/bin/bash
mkdir -p gpurun_out/r03w
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "rotary" > gpurun_out/r03w/pytest.log 2>&1; tail -2 gpurun_out/r03w/pytest.log
python scripts/sweep_shapes.py 1 12 > gpurun_out/r03w/sweep_b1f12.log 2>&1
python scripts/sweep_shapes.py 1 3 > gpurun_out/r03w/sweep_b1f3.log 2>&1
grep -v amdgpu.ids gpurun_out/r03w/sweep_b1f12.log | cut -c1-200
grep -v amdgpu.ids gpurun_out/r03w/sweep_b1f3.log | cut -c1-200
