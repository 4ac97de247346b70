#!/bin/bash
# same-box A/B of env settings on the STEP alone (no tables): ab_step.sh "" "VAR=1" ...; frames/s and step median, alternating, 3 reps
export GEECO_DEV=1
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 100 --warmup 20 --skip-cpu --skip-layers --skip-other-configs --skip-input-pipeline --skip-inference --skip-dp-one-rank > gpurun_out/ab/s.json 2>gpurun_out/ab/s.err || { tail -5 gpurun_out/ab/s.err; continue; }
  python - "$e" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab/s.json').read().strip().splitlines()[-1])
print('[%s] %.0f frames/s  ms/step %.4f  median %.4f  p10 %.4f p90 %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['step_ms']['p90']))
PY
done
done
