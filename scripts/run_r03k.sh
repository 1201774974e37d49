#!/bin/bash
# name the first differing op of the training-step co-tenant mismatches
mkdir -p gpurun_out/r03k
timeout 1700 python scripts/exp_flake_train.py --iters 2000 --trace 1 > gpurun_out/r03k/flake_train_trace.log 2>&1
cut -c1-400 gpurun_out/r03k/flake_train_trace.log | tail -n 80
