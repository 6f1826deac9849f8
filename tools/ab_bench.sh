#!/bin/bash
# A/B on ONE box: bench.py against package copies under build/<dir>/ (older / variant libstcd_hip.so) and against the tree.
#   tools/ab_bench.sh "ab ab1" --model segcd --steps 40 --warmup 10
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
DIRS=$1; shift
for round in 1 2; do
  for d in $DIRS tree; do
    if [ $d = tree ]; then P=$R; else P=build/$d; fi
    python3 -c "
import sys, runpy
sys.path.insert(0, '$P'); import stcd_amd, stcd_amd._lib
sys.argv = ['bench.py'] + '$* --no-cpu-baseline'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('$d', d['ms_per_step'], d['value'], {n: v for n, v in d['roofline']['kernel_ms_per_step'].items() if 'gemm' in n})" || exit 1
  done
done
