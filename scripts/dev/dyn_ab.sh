#!/bin/bash
# input stage alone (cold caches), product library and ablation builds of dynimg.hip (build_variant.sh _nosync -DDYN_ABL_NOSYNC ...)
export GEECO_DEV=1
for rep in 1 2; do
for v in "" "$@"; do
  GEECO_LIB=libgeeco_hip$v.so timeout -k 10 120 python scripts/dev/u8_input_bench.py 2>&1 | tail -1
done
done
