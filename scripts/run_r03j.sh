#!/bin/bash
# localise the training-step co-tenant mismatch: many repeats, per-tensor report
mkdir -p gpurun_out/r03j
timeout 1500 python scripts/exp_flake_train.py --iters 2500 --graph-first 1 > gpurun_out/r03j/flake_train.log 2>&1
tail -c 6000 gpurun_out/r03j/flake_train.log
