#!/bin/bash
# build/variants/libseer_<name>.so: the library with gemm.hip recompiled under other flags (A/B and bug-hunt builds; select with
# SEER_HIP_LIB=<path>).  usage: scripts/build_variant.sh <name> [--no-vgpr-form] [extra hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
VG="-mllvm -amdgpu-mfma-vgpr-form=1"
if [ "$1" == "--no-vgpr-form" ]; then VG=""; shift; fi
mkdir -p $ROOT/build/variants
OBJ=$ROOT/build/variants/gemm_$NAME.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/seervideoldm_amd/csrc -fno-gpu-rdc -Wno-unused-result $VG "$@" \
    -c $ROOT/seervideoldm_amd/csrc/gemm.hip -o $OBJ
OTHERS=$(ls $ROOT/seervideoldm_amd/lib/obj/*.o | grep -v "/gemm.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/libseer_$NAME.so $OBJ $OTHERS
echo $ROOT/build/variants/libseer_$NAME.so
