#!/bin/bash
# FETCH_SIZE of one conv layer's kernels under env settings. usage: pmc_layer.sh L VAR=val [VAR=val ...]
R=$GRAFT_REPO_ROOT
L=$1; shift
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  out=$R/gpurun_out/pmcl_${d//[^A-Za-z0-9_]/_}
  rm -rf $out; mkdir -p $out
  export $d
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o p -- python3 $R/scripts/dev/conv_bench.py $L 3 > $out/log.txt 2>&1 || { echo "rocprof failed"; tail -5 $out/log.txt; exit 1; }
  python3 - $out $d <<'PY'
import csv, sys, collections, re
d, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(d + '/p_counter_collection.csv')))
agg = collections.OrderedDict()
for r in rows:
  name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
  if not name.startswith('conv'): continue
  e = agg.setdefault((int(r['Dispatch_Id']), name), {})
  e[r['Counter_Name']] = float(r['Counter_Value'])
last = {}
for (i, name), e in agg.items(): last[name + ' grid=' + str(i % 1)] = (i, e)
seen = collections.OrderedDict()
for (i, name), e in agg.items(): seen[name] = e   # keeps the last dispatch of each kernel name
for name, e in seen.items():
  print('debug=%s %-44s rd %8.1f MB  wr %8.1f MB' % (tag, name[:44], e.get('FETCH_SIZE', 0) * 1024 * 2 / 1e6, e.get('WRITE_SIZE', 0) * 1024 / 1e6))
PY
done
