#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04c
python -m pytest tests/test_engine_gpu.py -k "virtual or every_layer" -x -q -s > gpurun_out/r04c/virt.log 2>&1; echo "virt rc=$?" | tee -a gpurun_out/r04c/rc.txt
python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/r04c/ops.log 2>&1; echo "ops rc=$?" | tee -a gpurun_out/r04c/rc.txt
for v in 1 0 1 0; do
  STCD_VIRT_ACT=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r04c/bench_virt$v.json 2>> gpurun_out/r04c/bench.err; echo "bench virt=$v rc=$?" | tee -a gpurun_out/r04c/rc.txt
  STCD_BENCH_TOP_KERNELS=12 python -c "import json;d=json.load(open('gpurun_out/r04c/bench_virt$v.json'));r=d['roofline'];print('virt=$v', d['value'], d['ms_per_step'], r['launches_per_step_all_kernels'], r['class_ms_per_step'], r['kernel_ms_per_step'])" | tee -a gpurun_out/r04c/ab.txt
done
for m in snunet segcd; do
  python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r04c/$m.json 2>> gpurun_out/r04c/bench.err
  python -c "import json;d=json.load(open('gpurun_out/r04c/$m.json'));print('$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04c/ab.txt
done
tail -n 3 gpurun_out/r04c/virt.log; tail -n 3 gpurun_out/r04c/ops.log
