#!/bin/bash
# Builds the CURRENT csrc tree as geeco_amd/libgeeco_hip<suffix>.so with extra compiler flags, for same-box A/B runs
# (GEECO_DEV=1 GEECO_LIB=libgeeco_hip<suffix>.so python bench.py ...).  usage: build_variant.sh _noskew -DFB_SKEW=0
# (the PRODUCT kernel set; with the development variants and switches: build_dev_lib.sh)
set -euo pipefail
SUF=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/geeco_amd/csrc
B=build$SUF
rm -rf $B && mkdir -p $B
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $*"
extra_flags() {      # as geeco_amd/csrc/build.sh; PLAIN=1: none (to A/B the per-file settings themselves)
  [ -n "${PLAIN:-}" ] && return
  case $1 in
    conv_gemm) echo "-mllvm -amdgpu-sched-strategy=max-ilp" ;;
    conv_halo|conv_dgrad_lds) echo "-mllvm -amdgpu-use-amdgpu-trackers=1" ;;
  esac
}
pids=()
for f in conv_gemm conv_halo conv_wgrad conv_wgrad_halo conv_dgrad_lds dynimg decoder misc; do
  /opt/rocm/bin/hipcc $FLAGS $(extra_flags $f) -c $f.hip -o $B/$f.o &
  pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -x hip -c errors.cpp -o $B/errors.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgeeco_hip$SUF.so $B/*.o
rm -rf $B
echo "built geeco_amd/libgeeco_hip$SUF.so ($*)"
