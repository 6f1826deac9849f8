#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_engine_gpu.py -k "virtual or every_layer" -q -s > $O/virt.log 2>&1; echo "virt rc=$?" | tee -a $O/rc.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  STCD_VIRT_ACT=$v rocprofv3 --kernel-trace --output-format csv -d $O/trace$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace$v.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $O/trace$v $O/timeline_virt$v.txt > /dev/null
  rm -rf $O/trace$v
done
cd $GRAFT_REPO_ROOT
tail -n 3 $O/virt.log; tail -n 2 $O/timeline_virt0.txt $O/timeline_virt1.txt
