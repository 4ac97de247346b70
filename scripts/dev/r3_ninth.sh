#!/bin/bash
export GEECO_DEV=1
GEECO_WGRAD_TALL=2 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu -k "wgrad or config2" 2>&1 | tail -3
bash scripts/dev/ab_env.sh "" "GEECO_WGRAD_TALL=2" 2>&1 | tail -4
