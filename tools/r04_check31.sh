#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -q -m gpu -x > gpurun_out/c31_tests.log 2>&1 || { tail -30 gpurun_out/c31_tests.log; exit 1; }
tail -2 gpurun_out/c31_tests.log
for i in 1 2; do for f in 0 1 2; do
STCD_SMALL_FAST=$f python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c31_f${f}_$i.json 2>> gpurun_out/c31_err.log
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c31_f*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'].get('k_conv_small<1, 5>'))
PY
