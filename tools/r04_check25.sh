#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
for i in 1 2; do for p in 2048 1024 512; do
STCD_BN_CHUNK_PIECES=$p python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c25_p${p}_$i.json 2>> gpurun_out/c25_err.log
done; done
STCD_BN_CHUNK_PIECES=1024 python bench.py --model snunet --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c25_snunet_p1024.json 2>> gpurun_out/c25_err.log
python bench.py --model snunet --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c25_snunet_p2048.json 2>> gpurun_out/c25_err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c25_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
STCD_BN_CHUNK_PIECES=1024 bash tools/steptrace.sh > gpurun_out/c25_trace.txt 2>&1
python3 tools/step_timeline.py gpurun_out/steptrace gpurun_out/c25_timeline.txt > /dev/null
grep -E 'k_bn_reduce' gpurun_out/c25_timeline.txt
