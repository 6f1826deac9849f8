#!/bin/bash
# skip-layer recompute: parity + A/B timing
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "recomputed_skip or every_layer_in_place or fcsiam or matches or race" > gpurun_out/c18_tests.log 2>&1 || { tail -30 gpurun_out/c18_tests.log; exit 1; }
tail -3 gpurun_out/c18_tests.log
for i in 1 2; do
STCD_NO_SKIP_RECOMPUTE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c18_stored_$i.json 2> gpurun_out/c18_err.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c18_recomp_$i.json 2>> gpurun_out/c18_err.log
done
python bench.py --model sub --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c18_sub.json 2>> gpurun_out/c18_err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c18_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('launches_per_step'))
PY
