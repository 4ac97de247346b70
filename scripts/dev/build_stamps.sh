#!/bin/bash
# Builds geeco_amd/libgeeco_hip_stamps.so = the library with -DGEECO_STAMPS (in-kernel s_memtime timelines for
# scripts/dev/stamps.py).  Dev only: the product build never executes a stamp.
set -euo pipefail
cd "$(dirname "$0")/../../geeco_amd/csrc"
T=$(mktemp -d)
for f in conv_gemm conv_halo conv_wgrad conv_wgrad_halo conv_dgrad_lds dynimg decoder misc; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DGEECO_STAMPS ${STAMP_FLAGS:-} -c $f.hip -o $T/$f.o &
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -x hip -c errors.cpp -o $T/errors.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgeeco_hip_stamps.so $T/*.o
rm -rf $T
echo "built $(realpath ../libgeeco_hip_stamps.so)"
