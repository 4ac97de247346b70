#!/bin/bash
# round 3, first GPU call: the new parity tests, the DP / bench tests, then the bench line
set -o pipefail
mkdir -p gpurun_out/r3a
python -m pytest tests/test_bench_shapes_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_bench_gpu.py -x -q -s -m gpu > gpurun_out/r3a/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -40 gpurun_out/r3a/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 100 --warmup 20 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"; tail -25 gpurun_out/r3a/bench.err
