#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "conv1 or fwd or config2 or config4 or golden or bits" 2>&1 | tail -3
bash scripts/dev/ab_env.sh "" "GEECO_C1_NO_PACK3=1" 2>&1 | tee gpurun_out/ab_c1pack.txt | tail -4
