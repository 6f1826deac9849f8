#!/usr/bin/env python3
"""Probe: one training step captured into a hipGraph (torch.cuda.CUDAGraph) vs eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stcd_amd import synth
from stcd_amd.losses import cross_entropy
from stcd_amd.modules import SiamUnet_diff, SNUNet_ECAM
from stcd_amd.optim import FlatAdamW

arch = sys.argv[1] if len(sys.argv) > 1 else "diff"
dev = torch.device("cuda:0")
torch.manual_seed(1337)
model = (SNUNet_ECAM(3, 2, dtype="bf16") if arch == "snunet" else SiamUnet_diff(3, 2, dtype="bf16")).to(dev).train()
opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
a, b, lab = synth.make_batch(16, 256, 256, seed=1337)
A, B, L = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(lab).to(dev)


def step():
    opt.zero_grad(set_to_none=False)
    loss = cross_entropy(model(A, B), L)
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
print("eager default stream: %.3f ms/step" % timeit(step, 20))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        step()
    print("eager side stream   : %.3f ms/step" % timeit(step, 20))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        loss = step()
    g.replay()
    print("graph replay        : %.3f ms/step  (loss %.5f)" % (timeit(g.replay, 20), loss.item()))
