#!/bin/bash
# in-step A/B of a Python-level development switch: env_ab.sh <pattern> VAR  (alternating: unset, VAR=1; GEECO_DEV=1 both times)
R=$GRAFT_REPO_ROOT
pat=$1; var=$2
export GEECO_DEV=1
for rep in 1 2; do
for v in off on; do
  if [ $v = on ]; then export $var=1; else unset $var; fi
  MODE=graph bash scripts/dev/step_prof.sh eab_$v > /dev/null 2>&1
  echo "[$var $v] $(grep -E "$pat" gpurun_out/eab_$v/step_trace.txt | grep '%' | awk '{printf "%s %s | ", $1, $4}') $(grep 'sum of kernel' gpurun_out/eab_$v/step_trace.txt)"
  find gpurun_out/eab_$v -name "*.csv" -delete
done
done
