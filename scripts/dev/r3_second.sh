#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3b
python -m pytest tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_bench_gpu.py "tests/test_bench_shapes_gpu.py::test_config2_every_conv_launch_elementwise" -x -q -s -m gpu > gpurun_out/r3b/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; grep -v "^$" gpurun_out/r3b/pytest.log | tail -30
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fused or bits or slab" > gpurun_out/r3b/pytest_k.log 2>&1; echo "kernels rc=$?"; tail -3 gpurun_out/r3b/pytest_k.log
./scripts/dev/ub/mfma_shape > gpurun_out/r3b/mfma_shape.txt 2>&1; cat gpurun_out/r3b/mfma_shape.txt
./scripts/dev/ub/mfma_lds > gpurun_out/r3b/mfma_lds.txt 2>&1; cat gpurun_out/r3b/mfma_lds.txt
bash scripts/dev/libsweep.sh "" _noskew 2>&1 | tee gpurun_out/r3b/libsweep.txt | cut -c1-400
timeout -k 10 500 python bench.py --steps 100 --warmup 20 > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err; echo "bench rc=$?"; tail -8 gpurun_out/r3b/bench.err
