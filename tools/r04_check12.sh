#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/$name.json 2>> $O/bench.err
  STCD_BENCH_TOP_KERNELS=14 python -c "import json;d=json.load(open('$O/$name.json'));r=d['roofline'];print('$name', d['value'], d['ms_per_step'], {k:v for k,v in r['kernel_ms_per_step'].items() if 'x4' in k or 'bias' in k})" | tee -a $O/ab.txt
}
run seq0 STCD_X4_SEQ=0 STCD_BENCH_TOP_KERNELS=30
run seq1 STCD_X4_SEQ=1 STCD_BENCH_TOP_KERNELS=30
run auto STCD_BENCH_TOP_KERNELS=30
run seq0b STCD_X4_SEQ=0 STCD_BENCH_TOP_KERNELS=30
run autob STCD_BENCH_TOP_KERNELS=30
for m in snunet conc; do for v in 0 1; do
  STCD_X4_SEQ=$v python bench.py --model $m --steps 15 --warmup 4 --no-cpu-baseline --no-roofline > $O/${m}_seq$v.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/${m}_seq$v.json'));print('$m seq$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done
python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -q -k "not side_stream" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.txt; tail -n 2 $O/tests.log
TAG=virt0 STCD_VIRT_ACT=0 bash tools/pmc_step.sh > $O/pmc_virt0.log 2>&1
TAG=virt1 STCD_VIRT_ACT=1 bash tools/pmc_step.sh > $O/pmc_virt1.log 2>&1
tail -n 6 $O/pmc_virt1.log
