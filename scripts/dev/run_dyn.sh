mkdir -p gpurun_out/r5g
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "dynimg or goal" > gpurun_out/r5g/t.log 2>&1; echo rc=$?; tail -3 gpurun_out/r5g/t.log
timeout -k 10 120 python scripts/dev/u8_input_bench.py 2>&1 | tail -1
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --skip-cpu --skip-other-configs --skip-input-pipeline --skip-inference --skip-dp-one-rank > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err; echo bench rc=$?
python -c "
import json; d=json.loads(open('gpurun_out/r5g/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']); [print(r) for r in d['hbm']]"
