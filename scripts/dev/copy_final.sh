#!/bin/bash
# copies the set round_final.sh left under gpurun_out/r5final into profiles/r05/final_* (run here, after the gpurun call)
set -e
S=gpurun_out/r5final; D=profiles/r05
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
