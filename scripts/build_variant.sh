#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFLAG=1 ..."   -> obvi-slam_amd/csrc/variants/libobvi_ba_NAME.so (for OBVI_BA_LIBRARY=... A/B runs)
set -e
cd "$(dirname "$0")/../obvi-slam_amd/csrc"
mkdir -p variants/$1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Wno-unused-function $2"
for f in ba_kernels chol_kernels select_kernels frontend_kernels peak_kernels plan_kernels; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o variants/$1/$f.o & done
for f in abi upload plan lm; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $f.cpp -o variants/$1/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libobvi_ba_$1.so variants/$1/*.o
echo variants/libobvi_ba_$1.so
