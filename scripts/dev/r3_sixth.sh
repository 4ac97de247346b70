#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3g
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_estimator_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "lstm_step or parity or golden or trajectory or estimator or ranks" > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3g/pytest.log | cut -c1-300
bash scripts/dev/ab_env.sh "" "GEECO_NO_LSTM_BATCH=1" 2>&1 | tee gpurun_out/r3g/ab.txt
