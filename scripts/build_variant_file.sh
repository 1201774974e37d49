#!/bin/bash
# build/variants/libseer_<name>.so from a source file OUTSIDE the tree in place of one library source (measurement builds that must
# not change the library's source digest).  usage: scripts/build_variant_file.sh <name> <replaced source.hip> <file> [hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; FILE=$3; shift; shift; shift
VG=""
case $SRC in gemm.hip|gemm_t320.hip|gemm_ws.hip|gemm_tn.hip) VG="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
mkdir -p $ROOT/build/variants
TMP=$ROOT/build/variants/${SRC%.hip}_$NAME.hip
cp $FILE $TMP
OBJ=$ROOT/build/variants/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/seervideoldm_amd/csrc -fno-gpu-rdc -Wno-unused-result \
    $VG "$@" -c $TMP -o $OBJ
OTHERS=$(ls $ROOT/seervideoldm_amd/lib/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/libseer_$NAME.so $OBJ $OTHERS
echo $ROOT/build/variants/libseer_$NAME.so
