#!/bin/bash
# Measurement build of gemm.hip with the in-kernel timeline stamps (-DSEER_GEMM_STAMPS) next to the real library, plus the
# harness that reads them.  Run here (CPU box):
#     bash scripts/probe_tilestamps.sh
# then on the GPU, e.g.:
#     build/lab_pp8stamps 6144 5120 640 1 0          # GEGLU projection of the 640-wide level, AUTO tile
#     build/lab_pp8stamps conv 24 32 320 320          # 3x3 conv of the 320-wide level
#     LAB_SPAN_DUMP=1 build/lab_pp8stamps ...         # + entry / exit of every block
set -e
cd "$(dirname "$0")/.."
python -m seervideoldm_amd.build >/dev/null
mkdir -p build/libprobe
cd seervideoldm_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -fno-gpu-rdc -Wno-unused-result \
    -mllvm -amdgpu-mfma-vgpr-form=1 -DSEER_GEMM_STAMPS -c csrc/gemm.hip -o ../build/libprobe/gemm_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/libprobe/libseer_hip.so ../build/libprobe/gemm_stamps.o \
    $(ls lib/obj/*.hip.o | grep -v "/gemm.hip.o")      # every other object of the library
cd ..
/opt/rocm/bin/hipcc -O2 -std=c++17 -Iinclude scripts/lab_pp8stamps.cpp -o build/lab_pp8stamps -Lbuild/libprobe -lseer_hip \
    -Wl,-rpath,'$ORIGIN/libprobe'
ls -la build/libprobe/libseer_hip.so build/lab_pp8stamps
