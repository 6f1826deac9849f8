#!/bin/bash
# PMC counters for one opbench case.  usage: KIND=conv|wgrad bash tools/pmc_op.sh <case> ; results summarised per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${KIND:-conv}
CASE=${1:-c128=32,32,32,128,128}
rm -rf $R/gpurun_out/pmcop; mkdir -p $R/gpurun_out/pmcop
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do      # separate passes: FETCH_SIZE takes 3 and WRITE_SIZE 2 of the 4 TCC counters (MI355X_MICROARCH.md); one
                                 # pass with both fails with "error code 38: Request exceeds the capabilities of the hardware to collect"
  i=$((i+1))
  OPBENCH_KIND=$K timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcop/s$i -- python3 $R/tools/opbench.py $CASE > $R/gpurun_out/pmcop/s$i.log 2>&1 || echo "set $i failed"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcop/s*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'stcd' not in n or 'pack' in n or 'reduce' in n: continue
        agg[n.split('(')[0][-28:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:36s} {sum(x)/len(x):16.0f}")
PY
