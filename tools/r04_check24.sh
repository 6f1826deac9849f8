#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "recomputed_skip or every_layer_in_place or fcsiam or matches or race or config1" > gpurun_out/c24_tests.log 2>&1 || { tail -30 gpurun_out/c24_tests.log; exit 1; }
tail -2 gpurun_out/c24_tests.log
for i in 1 2; do
STCD_NO_WGRAD_TAIL_SPLIT=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c24_nosplit_$i.json 2>> gpurun_out/c24_err.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c24_split_$i.json 2>> gpurun_out/c24_err.log
done
python bench.py --model conc --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c24_conc_split.json 2>> gpurun_out/c24_err.log
STCD_NO_WGRAD_TAIL_SPLIT=1 python bench.py --model conc --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c24_conc_nosplit.json 2>> gpurun_out/c24_err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c24_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
bash tools/steptrace.sh > gpurun_out/c24_trace.txt 2>&1
python3 tools/step_timeline.py gpurun_out/steptrace gpurun_out/c24_timeline.txt > /dev/null
tail -12 gpurun_out/c24_timeline.txt
