#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/$name.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/$name.json'));r=d['roofline'];print('$name', d['value'], d['ms_per_step'], r['launches_per_step_all_kernels'], {k:round(v,3) for k,v in r['class_ms_per_step'].items()})" | tee -a $O/ab.txt
}
V=$PWD/build/variants/libstcd_stash_after.so
run virt0_first STCD_VIRT_ACT=0
run virt0_after STCD_VIRT_ACT=0 STCD_LIB_PATH=$V
run virt1_m0 STCD_VIRT_ACT=1 STCD_XF_MODE=0
run virt1_m1 STCD_VIRT_ACT=1 STCD_XF_MODE=1
run virt1_m2 STCD_VIRT_ACT=1 STCD_XF_MODE=2
run virt1_m3 STCD_VIRT_ACT=1 STCD_XF_MODE=3
run virt1_m1_after STCD_VIRT_ACT=1 STCD_XF_MODE=1 STCD_LIB_PATH=$V
run virt1_m3_after STCD_VIRT_ACT=1 STCD_XF_MODE=3 STCD_LIB_PATH=$V
