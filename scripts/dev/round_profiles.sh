#!/bin/bash
# One GPU call that produces everything profiles/rNN needs: bench JSON, kernel stats + step trace (eager, no private
# roofline loops), PMC report, PMC traffic of the roofline kernel.  usage: round_profiles.sh <tag>
R=$GRAFT_REPO_ROOT
tag=${1:-prof}
out=$R/gpurun_out/$tag
rm -rf $out $R/gpurun_out/pmc; mkdir -p $out
cd $R
timeout -k 10 400 python bench.py --steps 100 --warmup 20 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
bash scripts/dev/step_prof.sh $tag/sprof > $out/step_prof.log 2>&1; tail -3 $out/step_prof.log
bash scripts/dev/pmc.sh > $out/pmc.log 2>&1; tail -5 $out/pmc.log
python3 scripts/dev/pmc_report.py $R/gpurun_out/pmc > $out/pmc.txt 2>&1; tail -3 $out/pmc.txt
K=$(python3 -c "import json;print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['roofline']['kernel'])")
python3 scripts/dev/pmc_roofline.py $R/gpurun_out/pmc "$K" $out/pmc_roofline_kernel.json
