#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3e
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_estimator_gpu.py tests/test_predictor_gpu.py tests/test_bench_gpu.py "tests/test_bench_shapes_gpu.py::test_config2_every_conv_launch_elementwise" "tests/test_full_size_gpu.py" -x -q -m gpu -k "not config4" > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3e/pytest.log | cut -c1-300
bash scripts/dev/ab_env.sh "" "GEECO_NO_CONV1_NORM=1" "GEECO_LIB=libgeeco_hip_c1contig.so" "GEECO_LIB=libgeeco_hip_c1contig.so GEECO_NO_CONV1_NORM=1" 2>&1 | tee gpurun_out/r3e/ab.txt
