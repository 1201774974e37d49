#!/bin/bash
# round 4: what bounds the main loop of the 256 x 320 tile kernel -- measurement builds with one resource removed at a time
# (SEER_T320_PROBE: 1 = no LDS-DMA inside the loop, 2 = no LDS fragment reads, 4 = no MFMAs; results are wrong by design),
# on two long-loop shapes at config-4 rows; then the vendor-library calibration with shapes the big tile takes
mkdir -p gpurun_out/r04
L=gpurun_out/r04/t320_probe.log
: > $L
for shape in "ff1 geglu L2" "conv 16x16 1920" "ff2 +res L1"; do
  echo "== $shape (config-4 rows), shipped library" >> $L
  LAB_ONLY="$shape" LAB_MMUL=4 timeout 120 build/lab_gemm 20 22 2>&1 | grep -E "^(ff|conv)" >> $L
  for p in 1 2 4 3 5 6; do
    echo "-- probe $p" >> $L
    LD_PRELOAD=build/variants/libseer_t320p$p.so LAB_ONLY="$shape" LAB_MMUL=4 timeout 120 build/lab_gemm 20 22 2>&1 | grep -E "^(ff|conv)" >> $L
  done
done
echo "== vendor calibration" >> $L
timeout 600 python scripts/bench_vendor_gemm.py >> $L 2>&1
cat $L
