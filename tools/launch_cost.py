#!/usr/bin/env python3
"""Per-launch cost of a dependent chain of tiny kernels: default (null) stream vs a side stream vs hipGraph replay."""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
big = torch.zeros(8 << 20, device=dev)


def chain(n):
    for _ in range(n):
        x.add_(1.0)


def timed(fn, n, stream=None):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record(stream)
    fn(n)
    b.record(stream)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n, host * 1e6 / n


N = 4000
chain(200)
print("default stream   : %.2f us/launch on device, %.2f us/launch host enqueue" % timed(chain, N))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(200)
    print("side stream      : %.2f us/launch on device, %.2f us/launch host enqueue" % timed(chain, N, s))
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    chain(10)
    with torch.cuda.graph(g, stream=s):
        chain(N)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s); g.replay(); b.record(s)
    torch.cuda.synchronize()
    print("hipGraph replay  : %.2f us/launch on device" % (a.elapsed_time(b) * 1e3 / N))


def chain_big(n):
    for _ in range(n):
        big.add_(1.0)


chain_big(20)
print("32 MB add_ default: %.2f us/launch on device, %.2f host" % timed(chain_big, 500))
with torch.cuda.stream(s):
    chain_big(20)
    print("32 MB add_ side   : %.2f us/launch on device, %.2f host" % timed(chain_big, 500, s))
