#!/bin/bash
# Builds the C-ABI shared library for gfx950 in-tree (geeco_amd/libgeeco_hip.so): the PRODUCT kernel set only -- no GEECO_*
# switch is compiled in and no kernel that only a switch could select (those live in scripts/dev/build_dev_lib.sh's library).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libgeeco_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -amdgpu-mfma-vgpr-form: accumulators stay in VGPRs (unified file on gfx950); without it the allocator parks them in AGPRs in
# some kernels and pays v_accvgpr_read/write copies, each of which costs MFMA issue time
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
# per-file scheduler settings (same-box A/B of whole-library variants, per-layer table: profiles/r03/ab_compiler_flags.txt):
# the gather GEMM gains 2-3 % from the max-ILP strategy (conv4-6 forward), the LDS-halo and LDS-staged input-gradient
# kernels 0.5-1 % from the AMDGPU register-pressure trackers; every other combination measured was neutral or worse
extra_flags() {
  case $1 in
    conv_gemm) echo "-mllvm -amdgpu-sched-strategy=max-ilp" ;;
    conv_halo|conv_dgrad_lds) echo "-mllvm -amdgpu-use-amdgpu-trackers=1" ;;
  esac
}
rm -rf build && mkdir -p build
pids=()
for f in conv_gemm conv_halo conv_wgrad conv_wgrad_halo conv_dgrad_lds dynimg decoder misc; do
  $HIPCC $FLAGS $(extra_flags $f) -c $f.hip -o build/$f.o &
  pids+=($!)
done
$HIPCC $FLAGS -x hip -c errors.cpp -o build/errors.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT build/*.o
g++ -O3 -std=c++17 -Wall -fPIC -shared -o ../libgeeco_host.so host_io.cpp host_inflate.cpp -lz -lpthread
echo "built $(realpath $OUT) and $(realpath ../libgeeco_host.so)"
