#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_cf_ops_gpu.py tests/test_engine_gpu.py tests/test_segcd_gpu.py -q -m gpu -x > gpurun_out/c32_tests.log 2>&1 || { tail -30 gpurun_out/c32_tests.log; exit 1; }
tail -2 gpurun_out/c32_tests.log
PRE=$GRAFT_REPO_ROOT/build/pre_vgpr/libstcd_hip_pre.so
for m in "diff" "conc" "snunet" "segcd" "changeformer" "changeformer --encoder mit_b0"; do
  tag=$(echo $m | tr -d ' -' )
  for i in 1 2; do
    STCD_LIB_PATH=$PRE python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c32_${tag}_pre_$i.json 2>> gpurun_out/c32_err.log
    python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c32_${tag}_new_$i.json 2>> gpurun_out/c32_err.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c32_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
