#!/bin/bash
export GEECO_DEV=1   # the product reads GEECO_* switches only under GEECO_DEV=1
# bench value under a list of env settings (same box, same build): envsweep.sh "A=1" "B=2 C=3" ...
for e in "" "$@"; do
  v=$(env $e python bench.py --steps 40 --warmup 8 --skip-cpu --skip-layers 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['step_ms']['median'])")
  echo "[$e] $v"
done
