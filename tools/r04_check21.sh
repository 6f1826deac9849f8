#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
bash tools/steptrace.sh > gpurun_out/c21_trace.txt 2>&1
python3 tools/step_timeline.py gpurun_out/steptrace gpurun_out/c21_timeline.txt > /dev/null
grep -E 'k_skip_bwd|k_bn_act_pair' gpurun_out/c21_timeline.txt
tail -3 gpurun_out/c21_timeline.txt
