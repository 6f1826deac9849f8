#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $O
python -m pytest tests/test_changeformer_gpu.py -k "mitb0 or emulating" -q -s > $O/cf.log 2>&1; echo "cf rc=$?" | tee -a $O/rc.txt
python -m pytest tests/test_bf16_emulation_gpu.py -k snunet -q -s > $O/sn_emul.log 2>&1; echo "sn_emul rc=$?" | tee -a $O/rc.txt
STCD_BENCH_TOP_KERNELS=24 python bench.py --model changeformer --encoder mit_b0 --steps 10 --warmup 3 > $O/bench_mitb0.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python -c "import json;d=json.load(open('$O/bench_mitb0.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], r['launches_per_step_all_kernels'], r['step']); print(r['kernel_ms_per_step'])"
grep -E "trained state|engine vs|emulation vs|classes|conv / attention|passed|failed|^E  " $O/cf.log $O/sn_emul.log | cut -c1-330 | tail -n 30
