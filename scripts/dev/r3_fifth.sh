#!/bin/bash
export GEECO_DEV=1
bash scripts/dev/step_prof.sh r3f_new > /dev/null 2>&1; head -8 gpurun_out/r3f_new/step_trace.txt; grep "sum of kernel" gpurun_out/r3f_new/step_trace.txt
GEECO_NO_CONV1_NORM=1 bash scripts/dev/step_prof.sh r3f_old > /dev/null 2>&1; head -8 gpurun_out/r3f_old/step_trace.txt; grep "sum of kernel" gpurun_out/r3f_old/step_trace.txt
