#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py tests/test_golden_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "wgrad or config2 or slab or golden or parity" 2>&1 | tail -3
bash scripts/dev/ab_env.sh "" "GEECO_LIB=libgeeco_hip_old.so" 2>&1 | tail -4
