#!/bin/bash
# round-4 judged artefacts: bench line + rocprofv3 kernel stats + PMC traffic for the headline and the other families; per-op PMC re-captures
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1
bash tools/profile_round.sh r04_snunet --model snunet >> gpurun_out/r04_profile.log 2>&1
bash tools/profile_round.sh r04_segcd --model segcd >> gpurun_out/r04_profile.log 2>&1
bash tools/profile_round.sh r04_conc --model conc >> gpurun_out/r04_profile.log 2>&1
bash tools/profile_round.sh r04_changeformer --model changeformer >> gpurun_out/r04_profile.log 2>&1
bash tools/profile_round.sh r04_mitb0 --model changeformer --encoder mit_b0 >> gpurun_out/r04_profile.log 2>&1
KIND=conv bash tools/pmc_op.sh c64b=32,128,128,64,64 > gpurun_out/r04_conv_res_c64_pmc.txt 2>&1
OPBENCH_IMPL=6 KIND=conv bash tools/pmc_op.sh cf512=4,512,512,256,256 > gpurun_out/r04_conv_dma_pmc.txt 2>&1
tail -n 3 gpurun_out/r04_profile.log; tail -n 12 gpurun_out/r04_conv_res_c64_pmc.txt
