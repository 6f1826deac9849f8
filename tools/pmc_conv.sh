cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OPBENCH_KIND=conv OPBENCH_IMPL=1 timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc2a -- python3 $R/tools/opbench.py c16=32,256,256,16,16 > $R/gpurun_out/pmc2a.log 2>&1
OPBENCH_KIND=conv OPBENCH_IMPL=1 timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2b -- python3 $R/tools/opbench.py c16=32,256,256,16,16 > $R/gpurun_out/pmc2b.log 2>&1
OPBENCH_KIND=conv OPBENCH_IMPL=1 timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc2c -- python3 $R/tools/opbench.py c16=32,256,256,16,16 > $R/gpurun_out/pmc2c.log 2>&1
OPBENCH_KIND=conv OPBENCH_IMPL=1 timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc2d -- python3 $R/tools/opbench.py c16=32,256,256,16,16 > $R/gpurun_out/pmc2d.log 2>&1
ls $R/gpurun_out/pmc2*/
