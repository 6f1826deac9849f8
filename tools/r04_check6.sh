#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O
python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.txt
run() { local name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/$name.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/$name.json'));r=d['roofline'];print('$name', d['value'], d['ms_per_step'], r['launches_per_step_all_kernels'], {k:round(v,3) for k,v in r['class_ms_per_step'].items()}, {k:v for k,v in r['kernel_ms_per_step'].items() if 'small' in k})" | tee -a $O/ab.txt
}
run virt0 STCD_VIRT_ACT=0
run virt1_m0 STCD_VIRT_ACT=1 STCD_XF_MODE=0
run virt1_m1 STCD_VIRT_ACT=1 STCD_XF_MODE=1
run virt0b STCD_VIRT_ACT=0
python bench.py --model snunet --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/snunet.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/snunet.json'));print('snunet', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
python bench.py --model segcd --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/segcd.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/segcd.json'));print('segcd', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
tail -n 3 $O/tests.log
