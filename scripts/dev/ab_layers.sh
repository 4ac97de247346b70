#!/bin/bash
# same-box A/B of library builds / env settings with the WHOLE per-layer table: ab_layers.sh "" "GEECO_LIB=libgeeco_hip_x.so" ...
# prints, per setting, the step median and every conv launch's us (median of 30 x 5 launches)
export GEECO_DEV=1
mkdir -p gpurun_out/ab
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 60 --warmup 10 --skip-cpu --skip-other-configs > gpurun_out/ab/l.json 2>gpurun_out/ab/l.err || { tail -5 gpurun_out/ab/l.err; continue; }
  python - "$e" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab/l.json').read().strip().splitlines()[-1])
print('[%s] step median %.4f ms' % (sys.argv[1], d['step_ms']['median']))
print('   ' + ' '.join('%s:%s=%.1f' % (r['layer'][4:], r['op'][:5], r['us']) for r in d['layers']))
PY
done
