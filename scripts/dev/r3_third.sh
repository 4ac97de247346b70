#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3c
python scripts/dev/timing_probe.py > gpurun_out/r3c/timing_probe.txt 2>&1; cat gpurun_out/r3c/timing_probe.txt
python -m pytest tests/test_dp_gpu.py tests/test_full_size_gpu.py -x -q -s -m gpu -k "late_bucket or config4" > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^$" gpurun_out/r3c/pytest.log | tail -6 | cut -c1-400
bash scripts/dev/round_profiles.sh r3c_prof > gpurun_out/r3c/round_profiles.log 2>&1; tail -12 gpurun_out/r3c/round_profiles.log
