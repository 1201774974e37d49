#!/bin/bash
# Build measurement-only variants of the attention kernel (SEER_ATTN_PROBE bits: 1 = no v_exp, 2 = no PV MFMAs,
# 4 = no QK^T MFMAs, 8 = K/V tile staged once) next to the real library, for scripts/probe_attn.py.
set -e
cd "$(dirname "$0")/.."
python -m seervideoldm_amd.build >/dev/null
cd seervideoldm_amd
for p in 1 2 4 6 8 9 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -fno-gpu-rdc -Wno-unused-result \
      -mllvm -amdgpu-mfma-vgpr-form=1 -DSEER_ATTN_PROBE=$p -c csrc/attention.hip -o lib/obj/attn_probe$p.o &
done
wait
for p in 1 2 4 6 8 9 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libseer_aprobe$p.so lib/obj/attn_probe$p.o \
      $(ls lib/obj/*.hip.o | grep -v "/attention.hip.o")   # every other object of the library: _lib.load() binds all symbols
done
ls lib/*.so
