#!/bin/bash
# config-5 shape (geeco-f rgbd K=32): input stage in the step + ms/step, product library against variants.  cfg5_ab.sh "" _d1
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ -n "$v" ]; then export GEECO_DEV=1 GEECO_LIB=libgeeco_hip$v.so; else unset GEECO_DEV GEECO_LIB; fi
  out=$R/gpurun_out/c5$v; rm -rf $out; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r1 -- python3 $R/bench.py --channels 4 --seq-len 32 --steps 10 --warmup 3 --skip-cpu --skip-layers --skip-other-configs --skip-input-pipeline --skip-inference --skip-dp-one-rank > $out/prof.log 2>&1)
  python3 scripts/dev/step_trace.py $(find $out -name "*kernel_trace.csv" | head -1) > $out/step_trace.txt
  echo "[$v] $(grep -E "dynimg" $out/step_trace.txt | grep '%' | awk '{printf "%s %s | ", $1, $4}') $(grep 'sum of kernel' $out/step_trace.txt)"
  find $out -name "*.csv" -delete
done
done
