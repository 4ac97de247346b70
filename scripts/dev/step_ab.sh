#!/bin/bash
# in-step A/B of library builds: graph-replayed step under rocprofv3 --kernel-trace, chosen kernels + step sum.  step_ab.sh <pattern> "" _nt ...
R=$GRAFT_REPO_ROOT
pat=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ -n "$v" ]; then export GEECO_DEV=1 GEECO_LIB=libgeeco_hip$v.so; else unset GEECO_DEV GEECO_LIB; fi
  MODE=graph bash scripts/dev/step_prof.sh sab$v > /dev/null 2>&1
  echo "[$v] $(grep -E "$pat" gpurun_out/sab$v/step_trace.txt | grep '%' | awk '{printf "%s %s | ", $1, $4}') $(grep 'sum of kernel' gpurun_out/sab$v/step_trace.txt)"
  find gpurun_out/sab$v -name "*.csv" -delete
done
done
