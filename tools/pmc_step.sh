#!/bin/bash
# SQ counters of the kernels of one bench.py configuration, summed per kernel name and step (chip totals).
#   usage: TAG=<name> [ENV...] bash tools/pmc_step.sh [bench.py arguments]      e.g.  TAG=virt1 STCD_VIRT_ACT=1 bash tools/pmc_step.sh
# Separate rocprofv3 passes per counter set (MI355X_MICROARCH.md: the SQ block takes 8 counters per pass; FETCH_SIZE / WRITE_SIZE need
# their own passes each); python3 directly after `--`.
R=$GRAFT_REPO_ROOT; T=${TAG:-step}; O=$R/gpurun_out/pmc_$T; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/s$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $O/s$i.log 2>&1 || echo "set $i failed"
done
cd $R
python3 - "$O" <<'PY'
import csv, glob, collections, sys, re
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + '/s*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'stcd' not in n: continue
        n = n.split('(')[0].replace('void stcd::', '').replace('stcd::', '')[:44]
        agg[n][r['Counter_Name']] += float(r['Counter_Value']); cnt[n][r['Counter_Name']] += 1
steps = 4.0
names = sorted(agg, key=lambda k: -agg[k].get('GRBM_GUI_ACTIVE', 0))
cols = ['GRBM_GUI_ACTIVE', 'SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'FETCH_SIZE', 'WRITE_SIZE']
out = ["# per step (sum over launches, chip totals; SQ *_CYCLES in units of 4 clocks; FETCH_SIZE / WRITE_SIZE raw counter units, see MI355X_MICROARCH.md)",
       "kernel".ljust(46) + " ".join(c[-14:].rjust(15) for c in cols)]
for n in names[:28]:
    out.append(n.ljust(46) + " ".join(f"{agg[n].get(c, 0) / steps:15.4g}" for c in cols))
open(O + '.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out[:14]))
PY
