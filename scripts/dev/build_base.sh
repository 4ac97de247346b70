#!/bin/bash
# Builds the csrc of a git revision (default HEAD) as geeco_amd/libgeeco_hip_base.so for same-box A/B runs
# (GEECO_LIB=libgeeco_hip_base.so python bench.py ...).
set -euo pipefail
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
mkdir -p $T/geeco_amd $T/include
git -C $ROOT archive $REV geeco_amd/csrc include | tar -x -C $T
cd $T/geeco_amd/csrc
for f in conv_gemm conv_halo conv_wgrad dynimg decoder misc; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c $f.hip -o $f.o &
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -x hip -c errors.cpp -o errors.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/geeco_amd/libgeeco_hip_base.so *.o
rm -rf $T
echo "built $ROOT/geeco_amd/libgeeco_hip_base.so from $REV"
