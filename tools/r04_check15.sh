#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04o; mkdir -p $O
R3=$PWD/build/r03/libstcd_hip_r03.so
for m in diff snunet segcd; do
  for lib in r04 r03 r04 r03; do
    if [ $lib = r03 ]; then export STCD_LIB_PATH=$R3; else unset STCD_LIB_PATH; fi
    a="--model $m"; [ $m = diff ] && a=""
    python bench.py $a --steps 30 --warmup 8 --no-cpu-baseline > $O/${m}_$lib.json 2>> $O/bench.err
    python -c "import json;d=json.load(open('$O/${m}_$lib.json'));r=d['roofline'];print('$m $lib', d['value'], d['ms_per_step'], {k:round(v,3) for k,v in r['class_ms_per_step'].items()})" | tee -a $O/ab.txt
  done
done
unset STCD_LIB_PATH
python -m pytest tests/test_ew_ops_gpu.py tests/test_engine_gpu.py -q -k "not side_stream" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.txt; tail -n 2 $O/tests.log
