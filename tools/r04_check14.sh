#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04n; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/$name.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/$name.json'));r=d['roofline'];print('$name', d['value'], d['ms_per_step'], round(r['class_ms_per_step']['conv'],3), {k:v for k,v in r['kernel_ms_per_step'].items() if 'conv_res' in k})" | tee -a $O/ab.txt
}
for t in 0 128 512 2048 0 512; do run one$t STCD_CONV_RES_ONE=$t STCD_BENCH_TOP_KERNELS=30; done
