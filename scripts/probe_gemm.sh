#!/bin/bash
# Build measurement-only variants of libseer_hip.so (SEER_GEMM_PROBE bits: 1 = no in-loop global->LDS refills,
# 2 = no LDS fragment reads, 4 = no MFMAs) next to the real library, for scripts/probe_gemm.py.  Run here (CPU box):
#     bash scripts/probe_gemm.sh
# then on the GPU:   python scripts/probe_gemm.py > gpurun_out/probe_gemm.log
set -e
cd "$(dirname "$0")/.."
python -m seervideoldm_amd.build >/dev/null
cd seervideoldm_amd
for p in 1 2 3 4 5 6 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -fno-gpu-rdc -Wno-unused-result \
      -DSEER_GEMM_PROBE=$p -c csrc/gemm.hip -o lib/obj/gemm_probe$p.o &
  if [ $((p % 4)) -eq 0 ]; then wait; fi
done
wait
for p in 1 2 3 4 5 6 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libseer_probe$p.so lib/obj/gemm_probe$p.o \
      $(ls lib/obj/*.hip.o | grep -v "/gemm.hip.o")      # every other object of the library: _lib.load() binds all symbols
done
ls -la lib/*.so
