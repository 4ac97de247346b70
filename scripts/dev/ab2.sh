#!/bin/bash
export GEECO_DEV=1   # the product reads GEECO_* switches only under GEECO_DEV=1
# A/B two env settings of bench.py (skip cpu leg)
for i in 1 2; do
  for e in "$@"; do
  echo -n "[$e] "; env $e python bench.py --steps 30 --warmup 5 --skip-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['encoder_forward']['tflops'])"
  done
done
