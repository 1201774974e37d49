#!/bin/bash
# build/variants/libseer_<name>.so: the library with ONE source recompiled under extra flags (measurement builds; use with
# LD_PRELOAD=<path> in front of a build/lab_* harness).  usage: scripts/build_variant_src.sh <name> <source.hip> [--agpr-form] [hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift; shift
VG="-mllvm -amdgpu-mfma-vgpr-form=1"
if [ "$1" == "--agpr-form" ]; then VG=""; shift; fi
mkdir -p $ROOT/build/variants
OBJ=$ROOT/build/variants/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/seervideoldm_amd/csrc -fno-gpu-rdc -Wno-unused-result \
    $VG "$@" -c $ROOT/seervideoldm_amd/csrc/$SRC -o $OBJ
OTHERS=$(ls $ROOT/seervideoldm_amd/lib/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/libseer_$NAME.so $OBJ $OTHERS
echo $ROOT/build/variants/libseer_$NAME.so
