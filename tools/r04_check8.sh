#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $O
R3=$PWD/build/r03/libstcd_hip_r03.so
for m in diff snunet segcd conc; do
  for lib in r04 r03 r04 r03; do
    if [ $lib = r03 ]; then export STCD_LIB_PATH=$R3; else unset STCD_LIB_PATH; fi
    a="--model $m"; [ $m = diff ] && a=""
    python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/${m}_$lib.json 2>> $O/bench.err
    python -c "import json;d=json.load(open('$O/${m}_$lib.json'));print('$m $lib', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
unset STCD_LIB_PATH
STCD_VIRT_ACT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/diff_virt1.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/diff_virt1.json'));print('diff r04 virt1', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_bf16_emulation_gpu.py -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.txt
tail -n 5 $O/tests.log
