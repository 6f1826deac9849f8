#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/$name.json 2>> $O/bench.err
  python -c "import json;d=json.load(open('$O/$name.json'));r=d['roofline'];print('$name', d['value'], d['ms_per_step'], {k:round(v,3) for k,v in r['class_ms_per_step'].items()})" | tee -a $O/ab.txt
}
run base STCD_SMALL_NT2=0
run nt2 STCD_SMALL_NT2=1
run base2 STCD_SMALL_NT2=0
run nt2b STCD_SMALL_NT2=1
STCD_SMALL_NT2=1 python -m pytest tests/test_engine_gpu.py -q -k "not side_stream" > $O/tests_nt2.log 2>&1; echo "tests nt2 rc=$?" | tee -a $O/rc.txt; tail -n 2 $O/tests_nt2.log
python -m pytest tests/test_changeformer_gpu.py -k "mitb0 or emulating" -q -s > $O/cf.log 2>&1; echo "cf rc=$?" | tee -a $O/rc.txt
grep -E "trained state|classes|passed|failed|^E  " $O/cf.log | cut -c1-300 | tail -n 12
