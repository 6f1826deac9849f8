#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O
python -m pytest tests/test_changeformer_gpu.py -k "emulating or eval_forward" -q -s > $O/cf_emul.log 2>&1; echo "cf_emul rc=$?" | tee -a $O/rc.txt
python -m pytest tests/test_bf16_emulation_gpu.py -k snunet -q -s > $O/sn_emul.log 2>&1; echo "sn_emul rc=$?" | tee -a $O/rc.txt
python -m pytest tests/test_ddp_gpu.py -q > $O/ddp.log 2>&1; echo "ddp rc=$?" | tee -a $O/rc.txt
python -m pytest tests/test_trainer_gpu.py -k "bit_reproducible or f1_parity" -q -s > $O/f1.log 2>&1; echo "f1 rc=$?" | tee -a $O/rc.txt
grep -E "trained state|engine vs|emulation vs|conv / attention|passed|failed" $O/cf_emul.log $O/sn_emul.log | tail -n 20
tail -n 3 $O/ddp.log; grep -E "bf16:|fp32:|passed|failed" $O/f1.log | tail -n 4
