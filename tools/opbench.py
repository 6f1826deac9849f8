#!/usr/bin/env python3
"""Time individual conv / wgrad launches of the engine through the C ABI (kernel-development aid).

  python tools/opbench.py [case ...]        cases: name=n,h,w,ci,co[,stride]   e.g. c128=32,32,32,128,128
Prints per-case MFMA-kernel time (HIP events, median of 20) with achieved TFLOP/s and GB/s.
  OPBENCH_KIND=conv|wgrad, OPBENCH_IMPL=<stcd_op_conv impl: 1 auto, 6 LDS-DMA ...>, OPBENCH_WIMPL=<stcd_op_wgrad impl: 1, 4, 7 LDS-DMA ...>
"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stcd_amd import _lib

DEV = "cuda:0"
IMPL = 1
TAPS3 = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
DEFAULT = ["c16=32,256,256,16,16", "c8=32,256,256,8,16", "c32=32,128,128,32,32", "c64=32,64,64,64,64", "c128=32,32,32,128,128",
           "d256=16,32,32,256,128", "d32=16,256,256,32,16"]


def geom(n, h, w, ci, co, ldo):
    g = _lib.ConvGeom()
    g.n, g.hi, g.wi, g.ci, g.ldi = n, h, w, ci, ci
    g.hm, g.wm, g.in_stride = h, w, 1
    g.ho, g.wo, g.out_stride, g.oy0, g.ox0 = h, w, 1, 0, 0
    g.co, g.ldo, g.ntaps = co, ldo, 9
    for i, (dy, dx) in enumerate(TAPS3):
        g.dy[i], g.dx[i] = dy, dx
    return g


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def main():
    l = _lib.lib()
    which = os.environ.get("OPBENCH_KIND", "conv,wgrad").split(",")
    global IMPL
    IMPL = int(os.environ.get("OPBENCH_IMPL", "1"))
    WIMPL = int(os.environ.get("OPBENCH_WIMPL", "1"))
    for case in (sys.argv[1:] or DEFAULT):
        name, spec = case.split("=")
        n, h, w, ci, co = [int(v) for v in spec.split(",")[:5]]
        ldo = (co + 7) // 8 * 8
        g = geom(n, h, w, ci, co, ldo)
        x = torch.randn(n, h, w, ci, device=DEV).bfloat16()
        wt = (torch.randn(9, ci, co, device=DEV) / np.sqrt(9 * ci)).float()
        out = torch.zeros(n, h, w, ldo, dtype=torch.bfloat16, device=DEV)
        dout = torch.randn(n, h, w, ldo, device=DEV).bfloat16()
        dw = torch.zeros(9, ci, co, device=DEV)
        nb = l.stcd_op_scratch_bytes(C.byref(g))
        scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        flops = 2.0 * n * h * w * 9 * ci * co
        byts = (n * h * w * (ci + co) + 9 * ci * co) * 2.0
        if "conv" in which:
            us = timeit(lambda: _lib.check(l.stcd_op_conv(1, IMPL, C.byref(g), C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), None,
                                                         C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), nb, st)))
            print(f"{name:6s} conv  {us:8.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  {byts / us / 1e3:8.1f} GB/s   (incl. filter repack launch)")
        if "wgrad" in which:
            us = timeit(lambda: _lib.check(l.stcd_op_wgrad(1, WIMPL, C.byref(g), C.c_void_p(x.data_ptr()), C.c_void_p(dout.data_ptr()),
                                                          C.c_void_p(dw.data_ptr()), C.c_void_p(scratch.data_ptr()), nb, st)))
            print(f"{name:6s} wgrad {us:8.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  {byts / us / 1e3:8.1f} GB/s   (incl. slab reduce launch)")


if __name__ == "__main__":
    main()
