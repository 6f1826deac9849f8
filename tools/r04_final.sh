#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee $O/rc.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python tools/show_bench.py $O/bench.json | head -8
python -m pytest tests/ -q -m gpu --durations=25 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/rc.txt
tail -n 6 $O/gpu_tests.log
