#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
for i in 1 2 3 4; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c41_off_$i.json 2>> gpurun_out/c41_err.log
  STCD_WGRAD_TAIL_SPLIT=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c41_on_$i.json 2>> gpurun_out/c41_err.log
done
python - <<'PY'
import json,glob,statistics
for k in ('off','on'):
    v=[json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'] for f in sorted(glob.glob(f'gpurun_out/c41_{k}_*.json'))]
    print(k, v, 'median', statistics.median(v))
PY
