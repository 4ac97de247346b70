#!/bin/bash
export GEECO_DEV=1   # the product reads GEECO_* switches only under GEECO_DEV=1
# same-box A/B of library builds: libsweep.sh "" _i0 ...  (suffixes of geeco_amd/libgeeco_hip<suffix>.so); prints the
# bench value and the first four rows of the per-layer table, twice per build (alternating)
mkdir -p gpurun_out/libsweep
for rep in 1 2; do
for v in "$@"; do
  GEECO_LIB=libgeeco_hip$v.so timeout -k 10 200 python bench.py --steps 30 --warmup 8 --skip-cpu --skip-other-configs --skip-input-pipeline --skip-inference --skip-dp-one-rank > gpurun_out/libsweep/b.json 2>gpurun_out/libsweep/b.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/libsweep/b.json').read().strip().splitlines()[-1])
print('[%s]' % sys.argv[1], d['value'], d['step_ms']['median'], [(r['layer'], r['op'], r['us']) for r in d["layers"]])
PY
done
done
