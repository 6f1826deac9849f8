#!/bin/bash
# round 4, second GPU check: virtual activations (forward staging transform) -- parity, then A/B bench
set -o pipefail
mkdir -p gpurun_out/r04b
python -m pytest tests/test_engine_gpu.py -k "virtual or every_layer or side_stream" -x -q -s > gpurun_out/r04b/virt.log 2>&1; echo "virt rc=$?" | tee -a gpurun_out/r04b/rc.txt
python -m pytest tests/test_engine_gpu.py tests/test_bf16_emulation_gpu.py tests/test_ddp_gpu.py -x -q > gpurun_out/r04b/engine.log 2>&1; echo "engine rc=$?" | tee -a gpurun_out/r04b/rc.txt
for v in 1 0 1 0; do
  STCD_VIRT_ACT=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r04b/bench_virt$v.json 2>> gpurun_out/r04b/bench.err; echo "bench virt=$v rc=$?" | tee -a gpurun_out/r04b/rc.txt
  python -c "import json;d=json.load(open('gpurun_out/r04b/bench_virt$v.json'));r=d['roofline'];print('virt=$v', d['value'], d['ms_per_step'], r['launches_per_step_all_kernels'], r['class_ms_per_step'])" | tee -a gpurun_out/r04b/ab.txt
done
for v in 1 0; do
  STCD_VIRT_ACT=$v python bench.py --model conc --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r04b/conc_virt$v.json 2>> gpurun_out/r04b/bench.err
  python -c "import json;d=json.load(open('gpurun_out/r04b/conc_virt$v.json'));print('conc virt=$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04b/ab.txt
done
python -m pytest tests/test_trainer_gpu.py -k "bit_reproducible or f1_parity" -x -q -s > gpurun_out/r04b/f1.log 2>&1; echo "f1 rc=$?" | tee -a gpurun_out/r04b/rc.txt
tail -n 4 gpurun_out/r04b/virt.log; tail -n 4 gpurun_out/r04b/engine.log; grep -E "bf16:|fp32:|passed|failed" gpurun_out/r04b/f1.log | tail -n 5
