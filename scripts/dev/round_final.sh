#!/bin/bash
# final measurement set of a round (usage: gpurun -- bash scripts/dev/round_final.sh; results under gpurun_out/r3final): default bench line, eager step trace + kernel stats, PMC report, roofline kernel traffic, roofline table
bash scripts/dev/round_profiles.sh r3final > gpurun_out/r3final_round_profiles.log 2>&1; tail -8 gpurun_out/r3final_round_profiles.log
python3 scripts/dev/roofline_table.py gpurun_out/r3final/bench.json gpurun_out/r3final/pmc.txt final > gpurun_out/r3final/roofline_table.md 2> gpurun_out/r3final/roofline_table.err; tail -3 gpurun_out/r3final/roofline_table.md
python bench.py --steps 20 --warmup 5 > gpurun_out/r3final/bench_driver_flags.json 2> gpurun_out/r3final/bench_driver_flags.err; echo "driver-flag bench rc=$?"
