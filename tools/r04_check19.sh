#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
bash tools/steptrace.sh > gpurun_out/c19_recomp.txt 2>&1
STCD_NO_SKIP_RECOMPUTE=1 bash tools/steptrace.sh > gpurun_out/c19_stored.txt 2>&1
grep -E 'skip_bwd|bn_act_pair|total' gpurun_out/c19_recomp.txt gpurun_out/c19_stored.txt
