#!/bin/bash
# kernel stats of the real-data step (u8 window addresses vs dense windows): scripts/dev/u8_step_time.py under rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT
tag=${1:-u8prof}
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r1 -- python3 $R/scripts/dev/u8_step_time.py > $out/prof.log 2>&1; echo "prof rc=$?"
cd $R
grep "ms/step" $out/prof.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $out/input_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
  n = r['Name']
  if any(k in n for k in ('dynimg', 'gather_windows', 'pack_pixels')):
    print('%-60s calls %6s  avg %9.1f us  min %9.1f us' % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
cat $out/input_kernels.txt
