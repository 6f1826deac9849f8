"""1x1-conv GEMM kernel over a list of shapes, 5 launches each, for a rocprofv3 kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gsweep -- python3 $R/tools/gemm_sweep.py
   python3 tools/gemm_sweep.py --parse gpurun_out/gsweep        # medians per shape, in launch order"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # n, h, w, ci, co
    (32, 8, 8, 64, 2048), (32, 8, 8, 128, 2048), (32, 8, 8, 256, 2048), (32, 8, 8, 512, 2048), (32, 8, 8, 1024, 2048), (32, 8, 8, 2048, 2048),
    (32, 8, 8, 2048, 512), (32, 16, 16, 1024, 256), (32, 16, 16, 256, 1024), (32, 32, 32, 512, 128), (32, 32, 32, 128, 512),
    (32, 64, 64, 64, 64), (32, 64, 64, 256, 64), (32, 64, 64, 64, 256), (32, 64, 64, 256, 128),
]
REP = 5
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_conv_gemm" in r["Kernel_Name"]]
    names = [r["Kernel_Name"] for r in rows if "k_conv_gemm" in r["Kernel_Name"]]
    for i, (n, h, w, ci, co) in enumerate(SHAPES):
        v = sorted(d[i * REP:(i + 1) * REP])
        us = v[len(v) // 2]
        fl = 2.0 * n * h * w * ci * co
        by = (n * h * w * (ci + co) + ci * co) * 2.0
        print(f"M={n*h*w:7d} K={ci:5d} N={co:5d}  {names[i*REP].split('(')[0][-14:]:14s} {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s  {by/us/1e3:7.0f} GB/s")
    sys.exit(0)
import ctypes as C
import numpy as np
import torch
from stcd_amd import _lib
l = _lib.lib()
DEV = "cuda:0"
for (n, h, w, ci, co) in SHAPES:
    g = _lib.ConvGeom()
    g.n, g.hi, g.wi, g.ci, g.ldi = n, h, w, ci, ci
    g.hm, g.wm, g.in_stride = h, w, 1
    g.ho, g.wo, g.out_stride, g.oy0, g.ox0 = h, w, 1, 0, 0
    g.co, g.ldo, g.ntaps = co, co, 1
    g.dy[0], g.dx[0] = 0, 0
    x = torch.randn(n, h, w, ci, device=DEV).bfloat16()
    wt = (torch.randn(1, ci, co, device=DEV) / np.sqrt(ci)).float()
    out = torch.zeros(n, h, w, co, dtype=torch.bfloat16, device=DEV)
    nb = l.stcd_op_scratch_bytes(C.byref(g))
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(REP):
        _lib.check(l.stcd_op_conv(1, 1, C.byref(g), C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), None, C.c_void_p(out.data_ptr()),
                                  C.c_void_p(scratch.data_ptr()), nb, st))
    torch.cuda.synchronize()
