#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04_full; mkdir -p $O
python -m pytest tests/ -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee $O/rc.txt
tail -n 15 $O/gpu_tests.log
