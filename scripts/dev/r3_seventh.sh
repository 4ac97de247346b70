#!/bin/bash
python scripts/dev/heads_stamps.py 2>/dev/null | tail -20
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "heads or parity or golden" 2>&1 | tail -3
bash scripts/dev/ab_env.sh "" "GEECO_HEADS_NO_LDS=1" 2>&1 | tail -4
