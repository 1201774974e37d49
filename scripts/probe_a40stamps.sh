#!/bin/bash
# measurement build: libseer_hip.so with attention40.hip compiled -DSEER_ATTN40_STAMPS into build/libprobe, + scripts/lab_a40stamps
set -e
cd "$(dirname "$0")/.."
python -m seervideoldm_amd.build >/dev/null
mkdir -p build/libprobe
cd seervideoldm_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -fno-gpu-rdc -Wno-unused-result \
    -mllvm -amdgpu-mfma-vgpr-form=1 -DSEER_ATTN40_STAMPS -c csrc/attention40.hip -o /tmp/attn40_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/libprobe/libseer_hip.so /tmp/attn40_stamps.o \
    $(ls lib/obj/*.hip.o | grep -v "/attention40.hip.o")
cd ..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude scripts/lab_a40stamps.cpp -o build/lab_a40stamps \
    -Lbuild/libprobe -lseer_hip -Wl,-rpath,'$ORIGIN/libprobe'
ls -la build/lab_a40stamps build/libprobe/
