#!/bin/bash
# Per-(kernel, grid) durations of bench.py steps: rocprofv3 --kernel-trace.   usage: bash tools/steptrace.sh [bench args]
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/steptrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/bench.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/steptrace/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
rows = list(csv.DictReader(open(f)))
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void stcd::', '').replace('void ', '')
    k = (name[:56], r.get('Grid_Size_X', r.get('Grid_Size', '?')))
    agg.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
steps = 13.0
tot = 0.0
out = []
for k, v in agg.items():
    s = sum(v) / steps
    tot += s
    out.append((s, k, len(v) / steps, sorted(v)[len(v) // 2]))
out.sort(reverse=True)
for s, k, n, med in out[:70]:
    print(f"{k[0]:56s} grid {k[1]:>9s}  per-step n={n:5.1f}  median {med:8.1f} us  sum/step {s:8.1f} us")
print(f"total kernel time per step {tot:.1f} us")
PY
tail -1 $OUT/bench.log
