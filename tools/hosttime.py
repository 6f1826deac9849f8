#!/usr/bin/env python3
"""Host enqueue time vs GPU time of one training step (is the step launch-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stcd_amd import synth
from stcd_amd.losses import cross_entropy
from stcd_amd.modules import SiamUnet_diff
from stcd_amd.optim import FlatAdamW

dev = "cuda:0"
m = SiamUnet_diff(3, 2, dtype="bf16").to(dev).train()
opt = FlatAdamW(m, lr=1e-3)
a, b, lab = synth.make_batch(16, 256, 256, seed=1)
A, B, L = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(lab).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = cross_entropy(m(A, B), L)
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/N:.3f} ms/step; total {1e3*(t2-t0)/N:.3f} ms/step")
# host-only cost with GPU idle between steps
ts = []
for _ in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
print(f"host enqueue (GPU idle) median {1e3*sorted(ts)[N//2]:.3f} ms/step")
