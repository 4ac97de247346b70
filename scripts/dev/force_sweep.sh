#!/bin/bash
# tile / split-K sweep of one gather-GEMM launch through GEECO_CONV_FORCE (development library): force_sweep.sh <layer> "<C:Nout:ncls>" "bm:bn:ks" ...
export GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev.so
L=$1; key=$2; shift 2
for rep in 1 2; do
for cfg in "" "$@"; do
  if [ -n "$cfg" ]; then export GEECO_CONV_FORCE="$key:$cfg"; else unset GEECO_CONV_FORCE; fi
  echo "[${cfg:-plan}] $(timeout -k 10 120 python scripts/dev/conv_bench.py $L 30 2>&1 | grep -E 'fwd' | tr -s ' ')"
done
done
