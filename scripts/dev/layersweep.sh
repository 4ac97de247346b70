#!/bin/bash
export GEECO_DEV=1   # the product reads GEECO_* switches only under GEECO_DEV=1
# per-layer times under a list of env settings (same box, same build): layersweep.sh "A=1" "B=2" ...; prints rows 0-4 of the layer table
mkdir -p gpurun_out/layersweep
for e in "" "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --skip-cpu > gpurun_out/layersweep/b.json 2>gpurun_out/layersweep/b.err
  python - "$e" <<'PY'
import json, os, sys
d = json.loads(open('gpurun_out/layersweep/b.json').read().strip().splitlines()[-1])
print('[%s]' % sys.argv[1], d['value'], d['step_ms']['median'], [(r['layer'], r['op'], r['us']) for r in d["layers"][int(os.environ.get("ROWS0", 4)):int(os.environ.get("ROWS1", 16))]])
PY
done
