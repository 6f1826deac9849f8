#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/$name.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/$name.json'));r=d['roofline'];print('$name', d['value'], d['ms_per_step'], round(r['class_ms_per_step']['conv'],3), {k:v for k,v in r['kernel_ms_per_step'].items() if 'conv_res' in k})" | tee -a $O/ab.txt
}
for kb in 38 76 38 76 112 150; do run kb$kb STCD_CONV_RES_KB=$kb STCD_BENCH_TOP_KERNELS=30; done
run nocw64 STCD_CONV_RES_NO_CW64=1 STCD_BENCH_TOP_KERNELS=30
run pipe1 STCD_CONV_RES_PIPE=1 STCD_BENCH_TOP_KERNELS=30
run gemm1 STCD_GEMM_OVER_RES=1 STCD_BENCH_TOP_KERNELS=30
