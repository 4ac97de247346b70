#!/bin/bash
# one GPU call: bench + rocprof kernel stats (+ optional tests)
set -o pipefail
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 2> gpurun_out/bench.err | tee gpurun_out/bench.json; echo "bench rc=$?"; tail -12 gpurun_out/bench.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python3 $R/bench.py --steps 10 --warmup 3 --skip-cpu --skip-other-configs --no-graph > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
cd $R; find gpurun_out/prof -name "*kernel_stats*" | head -3
