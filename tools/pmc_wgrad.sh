cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASES="c128=32,32,32,128,128 c16=32,256,256,16,16"
OPBENCH_KIND=wgrad timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc3a -- python3 $R/tools/opbench.py $CASES > $R/gpurun_out/pmc3a.log 2>&1
OPBENCH_KIND=wgrad timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc3b -- python3 $R/tools/opbench.py $CASES > $R/gpurun_out/pmc3b.log 2>&1
grep wgrad $R/gpurun_out/pmc3a.log
