#!/bin/bash
# rocprof kernel stats for an arbitrary bench configuration: prof_cfg.sh <tag> <bench args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o r -- python3 $R/bench.py --steps 4 --warmup 2 --skip-cpu --no-graph "$@" > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof rc=$?"
