#!/bin/bash
# kernel trace of a short bench run (eager launches; MODE=graph: hipGraph replay as in the real bench) -> per-step dispatch list (gpurun_out/$1/step_trace.txt) + stats csv
R=$GRAFT_REPO_ROOT
tag=${1:-sprof}
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r1 -- python3 $R/bench.py --steps 10 --warmup 3 --skip-cpu --skip-layers --skip-other-configs --skip-input-pipeline --skip-inference --skip-dp-one-rank $([ "$MODE" = graph ] || echo --no-graph) > $out/prof.log 2>&1; echo "prof rc=$?"
cd $R
python3 scripts/dev/step_trace.py $(find $out -name "*kernel_trace.csv" | head -1) > $out/step_trace.txt
tail -32 $out/step_trace.txt
