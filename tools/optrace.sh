#!/bin/bash
# Per-launch kernel durations of tools/opbench.py cases, grouped by (kernel, grid): rocprofv3 --kernel-trace.
# usage (on the GPU box): bash tools/optrace.sh [opbench cases...]
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/optrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/opbench.py "$@" > $OUT/opbench.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/optrace/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'].split('(')[0].replace('void stcd::', ''), r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
    agg.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k[0][:48]:48s} grid {k[1]:>8s},{k[2]:>3s},{k[3]:>3s}  n={len(v):3d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
cat $OUT/opbench.log
