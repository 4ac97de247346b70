#!/bin/bash
# one-rank data-parallel step (one graph, RCCL): wall per step from the kernel trace, a development switch off / on, alternating
# usage: dp_ab.sh VAR
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp GEECO_DEV=1
for rep in 1 2 3; do
for v in off on; do
  if [ $v = on ]; then export $1=1; else unset $1; fi
  rm -rf dpt; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d dpt -o r1 -- python3 $R/scripts/dev/dp_trace.py > dpt.log 2>&1
  echo "[$1 $v] $(python3 $R/scripts/dev/dp_trace_print.py $(find dpt -name '*kernel_trace.csv' | head -1) all | grep -E 'wgrad_reduce|dgrad_chunked|adam_seg|adam_kernel|step:' | awk '{printf "%s ", ($1 == "step:" ? "| step " $2 : $1)}')"
done
done
