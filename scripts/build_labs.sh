#!/bin/bash
# Build the stand-alone C++ harnesses under build/ (git-ignored; they travel to the GPU box with the gpurun snapshot).
set -e
cd "$(dirname "$0")/.."
python -m seervideoldm_amd.build >/dev/null
mkdir -p build
for lab in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude scripts/$lab.cpp -o build/$lab \
      -Lseervideoldm_amd/lib -lseer_hip -Wl,-rpath,'$ORIGIN/../seervideoldm_amd/lib'
done
ls -la build/
