#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04p; mkdir -p $O
R3=$PWD/build/r03/libstcd_hip_r03.so
run() { local name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/$name.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/$name.json'));r=d['roofline'];print('$name', d['value'], d['ms_per_step'], round(r['class_ms_per_step']['conv'],3), {k:v for k,v in r['kernel_ms_per_step'].items() if 'conv_res' in k})" | tee -a $O/ab.txt
}
for i in 1 2; do
run r03 STCD_LIB_PATH=$R3 STCD_BENCH_TOP_KERNELS=30
run d1 STCD_CONV_RES_TWO=0 STCD_BENCH_TOP_KERNELS=30
run two STCD_CONV_RES_TWO=1 STCD_BENCH_TOP_KERNELS=30
done
for m in snunet segcd; do for v in r03 0 1; do
  if [ $v = r03 ]; then e="STCD_LIB_PATH=$R3"; else e="STCD_CONV_RES_TWO=$v"; fi
  env $e python bench.py --model $m --steps 15 --warmup 4 --no-cpu-baseline --no-roofline > $O/${m}_$v.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/${m}_$v.json'));print('$m $v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done
STCD_CONV_RES_TWO=1 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -q -k "not side_stream" > $O/tests_two.log 2>&1; echo "tests two rc=$?" | tee -a $O/rc.txt; tail -n 2 $O/tests_two.log
python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -q -k "not side_stream" > $O/tests_d1.log 2>&1; echo "tests d1 rc=$?" | tee -a $O/rc.txt; tail -n 2 $O/tests_d1.log
