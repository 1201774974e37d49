#!/bin/bash
# round 4: wave priority in the 256 x 320 tile's ping-pong loop (SEER_T320_PRIO: 1 = raised while multiplying (shipped), 0 = none,
# -1 = raised while loading); the probes say LDS-DMA issue, fragment reads and MFMAs do not overlap at all
mkdir -p gpurun_out/r04
L=gpurun_out/r04/t320_prio.log
: > $L
for shape in "ff1 geglu L2" "conv 16x16 1920" "ff2 +res L1" "conv 32x32 320" "ff1 geglu L1"; do
  echo "== $shape (config-4 rows)" >> $L
  for v in shipped $@; do
    if [ $v == shipped ]; then PRE=""; else PRE=build/variants/libseer_$v.so; fi
    LD_PRELOAD=$PRE LAB_ONLY="$shape" LAB_MMUL=4 timeout 120 build/lab_gemm 20 22 2>&1 | grep -E "^(ff|conv)" | sed "s/^/$v  /" >> $L
  done
done
cat $L
