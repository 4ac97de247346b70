#!/bin/bash
# same-box A/B of the STEP time only: ab_step.sh REPS "ENV A" "ENV B" ... ; 100 graph-replayed steps per run, no layer table
export GEECO_DEV=1
reps=$1; shift
mkdir -p gpurun_out/ab
for rep in $(seq $reps); do
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 100 --warmup 20 --skip-cpu --skip-other-configs --skip-layers --skip-input-pipeline --skip-inference --skip-dp-one-rank > gpurun_out/ab/s.json 2>gpurun_out/ab/s.err || { tail -5 gpurun_out/ab/s.err; continue; }
  python - "$e" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab/s.json').read().strip().splitlines()[-1])
print('[%s] %.0f frames/s  ms_per_step %.4f  median %.4f  p10 %.4f  p90 %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['step_ms']['p90']))
PY
done
done
