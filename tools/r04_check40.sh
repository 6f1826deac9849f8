#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_bf16_emulation_gpu.py -q -m gpu -x > gpurun_out/c40_tests.log 2>&1 || { tail -30 gpurun_out/c40_tests.log; exit 1; }
tail -2 gpurun_out/c40_tests.log
for m in diff conc; do for i in 1 2 3; do
  STCD_BWDSUM_RES=0 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c40_${m}_small_$i.json 2>> gpurun_out/c40_err.log
  python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c40_${m}_res_$i.json 2>> gpurun_out/c40_err.log
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c40_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['class_ms_per_step']['conv'], r['class_ms_per_step']['bn_bwd_reduce'], r['launches_per_step_all_kernels'])
PY
