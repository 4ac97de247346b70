#!/bin/bash
# kernel-trace stats of single conv layers: layer_prof.sh L [L ...]   (results under gpurun_out/lprof_L/)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  out=$R/gpurun_out/lprof_$L
  rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $R/scripts/dev/conv_bench.py $L 10 > $out/log.txt 2>&1 || { echo "rocprof failed"; tail -5 $out/log.txt; exit 1; }
  grep "^conv" $out/log.txt
  python3 - $out <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
  print('   %-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
