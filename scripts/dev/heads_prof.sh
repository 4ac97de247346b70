#!/bin/bash
# kernel durations of the decoder tail alone (scripts/dev/heads_bench.py) under rocprofv3
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hb
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/hb -o r1 -- python3 $R/scripts/dev/heads_bench.py > $R/gpurun_out/hb/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, statistics, re
f = glob.glob("gpurun_out/hb/**/*kernel_trace.csv", recursive=True)[0]
per = {}
for r in csv.DictReader(open(f)):
  n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
  per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
  if "heads" in k: print(k, "median %.1f us min %.1f n %d" % (statistics.median(v), min(v), len(v)))
PY
