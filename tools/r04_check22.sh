#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "recomputed_skip or every_layer_in_place or fcsiam or matches or race" > gpurun_out/c22_tests.log 2>&1 || { tail -30 gpurun_out/c22_tests.log; exit 1; }
tail -2 gpurun_out/c22_tests.log
for v in 4 8; do
STCD_SKIP_PAIR_V=$v bash tools/steptrace.sh > gpurun_out/c22_trace_v$v.txt 2>&1
python3 tools/step_timeline.py gpurun_out/steptrace gpurun_out/c22_timeline_v$v.txt > /dev/null
echo "V=$v"; grep -E 'k_skip_bwd' gpurun_out/c22_timeline_v$v.txt; tail -1 gpurun_out/c22_timeline_v$v.txt
done
for i in 1 2; do for v in 4 8; do
STCD_SKIP_PAIR_V=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c22_v${v}_$i.json 2>> gpurun_out/c22_err.log
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c22_v*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
