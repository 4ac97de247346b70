#!/bin/bash
export GEECO_DEV=1   # the product reads GEECO_* switches only under GEECO_DEV=1
for a in "--model e2e_vmc --batch 64" "--channels 4 --seq-len 32" "--steps 20 --warmup 0"; do
  timeout -k 10 300 python bench.py $a --steps 10 --warmup 2 --skip-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['encoder_forward'])"
done
