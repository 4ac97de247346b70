#!/bin/bash
# copies the set round_final.sh left under gpurun_out/<tag> into profiles/<round>/final_* (run here, after the gpurun call)
#   usage: copy_final.sh [tag = r6final] [round = r06]
set -e
S=gpurun_out/${1:-r6final}; D=profiles/${2:-r06}; mkdir -p $D
cp $S/bench.json $D/final_bench.json
cp $S/bench_driver_flags.json $D/final_bench_driver_flags.json
cp $S/sprof/r1_kernel_stats.csv $D/final_kernel_stats.csv
cp $S/sprof/step_trace.txt $D/final_step_trace_eager.txt
cp $S/gprof/step_trace.txt $D/final_step_trace.txt
cp $S/kernel_stats_123_graph_launches.csv $D/final_kernel_stats_123_graph_launches.csv
cp $S/pmc.txt $D/final_pmc.txt
cp $S/roofline_table.md $D/final_roofline_table.md
cp $S/pmc_roofline_kernel.json $D/pmc_roofline_kernel.json
ls -la $D/final_* $D/pmc_roofline_kernel.json
