#!/bin/bash
# round 4: phase timeline of the 256 x 320 tile's main loop (measurement build -DSEER_T320_STAMPS): per phase of K tiles 4..6,
# ticks of 10 ns at: reads + LDS-DMA issued | fragment reads landed (lgkmcnt 0) | first barrier passed | MFMAs issued | second barrier passed
mkdir -p gpurun_out/r04
L=gpurun_out/r04/t320_stamps.log
LD_PRELOAD=build/variants/libseer_t320stamps.so LAB_STAMPS=1 LAB_STAMPS_ABS=1 LAB_SPLITS=1 LAB_ONLY="ff2 +res L1" LAB_MMUL=4 timeout 120 build/lab_gemm 5 22 > $L 2>&1
cat $L
