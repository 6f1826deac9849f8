#!/bin/bash
# round 4, first GPU check: deterministic reductions (race screens as bit-equality, reproducible training run, F1 statistic), bench line
set -o pipefail
mkdir -p gpurun_out/r04a
python -m pytest tests/test_engine_gpu.py -k "side_stream" -x -q -s > gpurun_out/r04a/side.log 2>&1; echo "side rc=$?" | tee -a gpurun_out/r04a/rc.txt
python -m pytest tests/test_trainer_gpu.py -k "bit_reproducible or f1_parity" -x -q -s > gpurun_out/r04a/f1.log 2>&1; echo "f1 rc=$?" | tee -a gpurun_out/r04a/rc.txt
python -m pytest tests/test_segcd_gpu.py -k "full_size" tests/test_changeformer_gpu.py -k "full_size or eval_forward" -x -q -s > gpurun_out/r04a/misc.log 2>&1; echo "misc rc=$?" | tee -a gpurun_out/r04a/rc.txt
python -m pytest tests/test_ops_gpu.py tests/test_ew_ops_gpu.py -x -q > gpurun_out/r04a/ops.log 2>&1; echo "ops rc=$?" | tee -a gpurun_out/r04a/rc.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r04a/rc.txt
python bench.py --model snunet --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04a/bench_snunet.json 2>> gpurun_out/r04a/bench.err; echo "snunet rc=$?" | tee -a gpurun_out/r04a/rc.txt
tail -3 gpurun_out/r04a/side.log gpurun_out/r04a/f1.log gpurun_out/r04a/misc.log gpurun_out/r04a/ops.log
