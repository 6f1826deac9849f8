"""Inference throughput (eval-mode forward, bf16, weights frozen so the filters are packed once): pairs/s for one family.
python tools/infer_bench.py --model diff|conc|snunet|segcd [--batch 16] [--size 256]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stcd_amd import synth
ap = argparse.ArgumentParser(); ap.add_argument("--model", default="diff"); ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256); ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
dev = "cuda:0"
if a.model == "segcd":
    from stcd_amd.segcd import SegCD
    m = SegCD().to(dev).eval()
else:
    from stcd_amd import modules
    m = {"diff": modules.SiamUnet_diff, "conc": modules.SiamUnet_conc, "snunet": modules.SNUNet_ECAM}[a.model](3, 2).to(dev).eval()
x1, x2, _ = synth.make_batch(a.batch, a.size, a.size, seed=5)
A, B = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)
for frozen in (False, True):
    import contextlib
    with torch.no_grad(), (m.frozen_weights() if frozen else contextlib.nullcontext()):
        for _ in range(5):
            m(A, B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            m(A, B)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print(f"{a.model} eval forward, {a.batch} pairs {a.size}x{a.size}, frozen_weights={frozen}: {dt * 1e3:.3f} ms  {a.batch / dt:.0f} pairs/s")
