#!/bin/bash
# Builds the DEVELOPMENT library geeco_amd/libgeeco_hip_dev<suffix>.so: the product sources with -DGEECO_DEV_KERNELS (every
# kernel variant a GEECO_* switch can select, the switches themselves: scripts/dev/SWITCHES.md) plus the development-only
# kernels under scripts/dev/experiments/ (conv_bottom_fwd).  The product library (geeco_amd/csrc/build.sh) contains none of it.
#   usage: build_dev_lib.sh [suffix] [-DFLAG ...]      then: GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev<suffix>.so python ...
set -euo pipefail
SUF=""
if [ $# -gt 0 ] && [[ "$1" != -* ]]; then SUF=$1; shift; fi
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
EXP=$ROOT/scripts/dev/experiments/conv_bottom_fwd
cd $ROOT/geeco_amd/csrc
B=build_dev$SUF
rm -rf $B && mkdir -p $B
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DGEECO_DEV_KERNELS -I$ROOT/geeco_amd/csrc -I$EXP $*"
extra_flags() {      # as geeco_amd/csrc/build.sh
  case $1 in
    conv_gemm) echo "-mllvm -amdgpu-sched-strategy=max-ilp" ;;
    conv_halo|conv_dgrad_lds|conv_bottom_fwd) echo "-mllvm -amdgpu-use-amdgpu-trackers=1" ;;
  esac
}
pids=()
for f in conv_gemm conv_halo conv_wgrad conv_wgrad_halo conv_dgrad_lds dynimg decoder misc; do
  /opt/rocm/bin/hipcc $FLAGS $(extra_flags $f) -c $f.hip -o $B/$f.o &
  pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS $(extra_flags conv_bottom_fwd) -c $EXP/conv_bottom_fwd.hip -o $B/conv_bottom_fwd.o &
pids+=($!)
/opt/rocm/bin/hipcc $FLAGS -x hip -c errors.cpp -o $B/errors.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgeeco_hip_dev$SUF.so $B/*.o
rm -rf $B
echo "built geeco_amd/libgeeco_hip_dev$SUF.so (-DGEECO_DEV_KERNELS $*)"
