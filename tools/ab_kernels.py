#!/usr/bin/env python3
"""Per-kernel A/B on one box: instrumented steps (stcd_profile_*) of one model through the package copy under build/ab/
(older library) and through the tree, each in its own process.   python tools/ab_kernels.py --model segcd [--encoder resnet50]"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time
sys.path.insert(0, sys.argv[1])
import torch
from stcd_amd import synth
from stcd_amd.losses import bce_dice_with_logits, cross_entropy
from stcd_amd.optim import FlatAdamW
model_name, encoder = sys.argv[2], sys.argv[3]
import os
if os.environ.get("AB_SEED"): torch.manual_seed(int(os.environ["AB_SEED"]))
if model_name == "segcd":
    from stcd_amd.segcd import SegCD
    m = SegCD(encoder_name=encoder, dtype="bf16")
else:
    from stcd_amd import modules
    m = {"diff": modules.SiamUnet_diff, "conc": modules.SiamUnet_conc, "snunet": modules.SNUNet_ECAM}[model_name](3, 2, dtype="bf16")
m = m.to("cuda:0").train()
opt = FlatAdamW(m, lr=1e-3)
a, b, lab = synth.make_batch(16, 256, 256, seed=1337)
A, B, L = (torch.from_numpy(v).cuda() for v in (a, b, lab))
def step():
    opt.zero_grad(set_to_none=True)
    o = m(A, B); o = o[-1] if isinstance(o, (list, tuple)) else o
    loss = bce_dice_with_logits(o, L.float().unsqueeze(1)) if o.shape[1] == 1 else cross_entropy(o, L)
    loss.backward(); opt.step()
for _ in range(8): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 30 * 1e3
eng = m._engine
eng.profile_enable(True)
for _ in range(3): step()
torch.cuda.synchronize()
k = eng.profile_kernels(); eng.profile_enable(False)
print(json.dumps({"wall_ms": wall, "kernels": {n: [v["ms"] / 3, v["launches"] // 3] for n, v in k.items()}}))
'''
args = sys.argv[1:]
model = args[args.index("--model") + 1] if "--model" in args else "segcd"
enc = args[args.index("--encoder") + 1] if "--encoder" in args else "resnet50"
res = {}
for tag, root in (("old", os.path.join(REPO, "build", "ab")), ("new", REPO)):
    out = subprocess.run([sys.executable, "-c", CHILD, root, model, enc], capture_output=True, text=True, cwd=REPO)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(out.stderr[-2000:]); sys.exit(1)
    res[tag] = json.loads(line[-1])
print(f"wall ms/step: old {res['old']['wall_ms']:.3f}  new {res['new']['wall_ms']:.3f}")
names = sorted(set(res["old"]["kernels"]) | set(res["new"]["kernels"]), key=lambda n: -res["new"]["kernels"].get(n, [0, 0])[0])
so = sn = 0.0
for n in names:
    o, w = res["old"]["kernels"].get(n, [0, 0]), res["new"]["kernels"].get(n, [0, 0])
    so += o[0]; sn += w[0]
    if abs(o[0] - w[0]) > 0.01 or w[0] > (0.3 if "--all" not in sys.argv else 0.0):
        print(f"{n:42s} old {o[0]:7.3f} ms ({o[1]:3d})   new {w[0]:7.3f} ms ({w[1]:3d})   {w[0] - o[0]:+.3f}")
print(f"sum of instrumented kernels: old {so:.3f}  new {sn:.3f}")
