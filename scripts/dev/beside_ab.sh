#!/bin/bash
# A/B of the optimiser beside the encoder bottom (development library): which piece, how wide.  usage: gpurun -- bash scripts/dev/beside_ab.sh
export GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev.so
mkdir -p gpurun_out/beside
for cfg in "both 0 2048" "both 512 512" "both 256 256" "both 1024 1024" "both 128 256" "adam 0 512" "adam 0 256" "reduce 256 2048" "reduce 512 2048"; do
  set -- $cfg
  echo "== GEECO_BESIDE=$1 GEECO_REDUCE_CAP=$2 GEECO_ADAM_CAP=$3"
  GEECO_BESIDE=$1 GEECO_REDUCE_CAP=$2 GEECO_ADAM_CAP=$3 timeout -k 10 200 python scripts/dev/beside_bottom_check.py 32 graph 2>&1 | grep "graph=True" | sed 's/.*ms.step/ms\/step/'
done
