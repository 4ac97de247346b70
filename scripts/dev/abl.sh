#!/bin/bash
for d in 0 1 2 3 4 6 7; do
  echo -n "debug=$d: "; GEECO_HALO_DEBUG=$d python bench.py --steps 5 --warmup 2 --skip-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], d['roofline']['achieved'])"
done
