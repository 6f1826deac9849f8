#!/usr/bin/env python3
"""Host-side enqueue cost: how long the CPU needs to enqueue one forward / one backward of a model (no synchronisation inside
the timed loop), per launch, beside the GPU time of the same calls.   python tools/host_cost.py --model segcd"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stcd_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="segcd")
ap.add_argument("--encoder", default="resnet50")
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
if a.model == "segcd":
    from stcd_amd.segcd import SegCD
    m = SegCD(encoder_name=a.encoder, dtype="bf16")
else:
    from stcd_amd import modules
    m = {"diff": modules.SiamUnet_diff, "conc": modules.SiamUnet_conc, "snunet": modules.SNUNet_ECAM}[a.model](3, 2, dtype="bf16")
m = m.to("cuda:0").train()
x1, x2, lab = (torch.from_numpy(v).cuda() for v in synth.make_batch(a.batch, 256, 256, seed=1))
eng = m._engine
for _ in range(3):
    o = m(x1, x2); o = o[-1] if isinstance(o, (list, tuple)) else o
    o.float().mean().backward()
torch.cuda.synchronize()
eng.profile_enable(True)
o = m(x1, x2); o = o[-1] if isinstance(o, (list, tuple)) else o
o.float().mean().backward()
torch.cuda.synchronize()
launches = sum(v["launches"] for v in eng.profile_kernels().values())
eng.profile_enable(False)
N = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    o = m(x1, x2); o = o[-1] if isinstance(o, (list, tuple)) else o
    o.float().mean().backward()
host = (time.perf_counter() - t0) / N
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / N
# one fwd+bwd at a time into an EMPTY queue (the loop above can fill the hardware queue: the host then waits for the GPU)
one = []
for _ in range(8):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    o = m(x1, x2); o = o[-1] if isinstance(o, (list, tuple)) else o
    t2 = time.perf_counter()
    o.float().mean().backward()
    one.append((t2 - t1, time.perf_counter() - t2))
torch.cuda.synchronize()
fwd1, bwd1 = min(v[0] for v in one), min(v[1] for v in one)
# a trivial torch kernel for scale
z = torch.zeros(64, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000):
    z.add_(1.0)
triv = (time.perf_counter() - t0) / 2000
torch.cuda.synchronize()
print(f"{a.model}: {launches} instrumented launches per fwd+bwd; host enqueue {host * 1e3:.3f} ms ({host / launches * 1e6:.1f} us per launch), "
      f"wall {wall * 1e3:.3f} ms; into an empty queue: forward {fwd1 * 1e3:.3f} ms + backward {bwd1 * 1e3:.3f} ms "
      f"({(fwd1 + bwd1) / launches * 1e6:.1f} us per launch); a trivial torch op costs the host {triv * 1e6:.1f} us")
