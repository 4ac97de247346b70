#!/bin/bash
# same-box A/B of env / library settings: ab_env.sh "VAR=1 VAR2=x" "GEECO_LIB=libgeeco_hip_x.so" ... ("" = defaults); prints
# frames/s, median step ms, conv1 fwd us and the in-step input-stage row, twice per setting (alternating)
export GEECO_DEV=1   # the product reads GEECO_* switches only under GEECO_DEV=1
mkdir -p gpurun_out/ab
for rep in 1 2; do
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 50 --warmup 10 --skip-cpu --skip-other-configs > gpurun_out/ab/b.json 2>gpurun_out/ab/b.err || { tail -5 gpurun_out/ab/b.err; continue; }
  python - "$e" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab/b.json').read().strip().splitlines()[-1])
rows = {(r['layer'], r['op']): r['us'] for r in d['layers']}
hb = [r for r in d['hbm'] if r['piece'].startswith('goal inputs as in the step')]
print('[%s] %.0f frames/s  step median %.4f ms  conv1 fwd %.1f us  conv2 fwd %.1f  fused bottom %.1f  input stage %s us  wgrad conv3..6 %s' % (
    sys.argv[1], d['value'], d['step_ms']['median'], rows[('conv1', 'fwd')], rows[('conv2', 'fwd')], rows[('conv2', 'dgrad+conv1_wgrad')],
    hb[0]['us'] if hb else '-', [rows[('conv%d' % l, 'wgrad')] for l in (3, 4, 5, 6)]))
import os
extra = os.environ.get('AB_ROWS', '').split()      # AB_ROWS="conv6:dgrad conv6:wgrad": further rows of the layer table
if extra:
  print('    ' + '  '.join('%s %.1f' % (e, rows[tuple(e.split(':'))]) for e in extra))
PY
done
done
