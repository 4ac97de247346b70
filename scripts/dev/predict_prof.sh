#!/bin/bash
R=$GRAFT_REPO_ROOT
for mdl in goal e2e; do
mkdir -p $R/gpurun_out/pt_$mdl
(cd /tmp && export TMPDIR=/tmp && PT_MODEL=$mdl rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pt_$mdl -o r1 -- python3 $R/scripts/dev/predict_trace.py > $R/gpurun_out/pt_$mdl/log.txt 2>&1)
python3 - $R/gpurun_out/pt_$mdl <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "") for r in rows]
# the last forward: from the last input-stage / pack kernel on
starts = [i for i, n in enumerate(names) if n.startswith('dynimg_goal') or (n.startswith('pack_pixels') and (i == 0 or not names[i-1].startswith('pack_pixels')))]
a = starts[-1]
t0 = int(rows[a]['Start_Timestamp']); tot = 0
agg = {}
for r, n in zip(rows[a:], names[a:]):
  d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
  agg[n[:50]] = agg.get(n[:50], [0, 0.0]); agg[n[:50]][0] += 1; agg[n[:50]][1] += d
print(sys.argv[1].split('_')[-1], 'forward: %d launches, kernel time %.1f us, span %.1f us' % (len(rows) - a, tot, (int(rows[-1]['End_Timestamp']) - t0) / 1e3))
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]: print('  %3d x %-50s %7.1f us' % (c, k, d))
PY
done
