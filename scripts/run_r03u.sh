#!/bin/bash
mkdir -p gpurun_out/r03u
python scripts/sweep_shapes.py 2 12 > gpurun_out/r03u/sweep_b2f12.log 2>&1
grep -v amdgpu.ids gpurun_out/r03u/sweep_b2f12.log | cut -c1-230
