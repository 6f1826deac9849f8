"""Per-kernel table of one training step through the engine's own HIP-event instrumentation (stcd_profile_*): launches,
ms, algorithmic TFLOP/s and GB/s per kernel name.  python tools/kernel_table.py --model snunet|segcd|diff|conc [--batch 16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stcd_amd import synth
from stcd_amd.losses import bce_dice_with_logits, cross_entropy

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="snunet")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
a = ap.parse_args()
dev = "cuda:0"
if a.model == "segcd":
    from stcd_amd.segcd import SegCD
    m = SegCD().to(dev).train()
    label = 1
else:
    from stcd_amd import modules
    cls = {"diff": modules.SiamUnet_diff, "conc": modules.SiamUnet_conc, "sub": modules.SiamUnet_sub, "snunet": modules.SNUNet_ECAM}[a.model]
    m = cls(3, 2).to(dev).train()
    label = 2
x1, x2, lab = synth.make_batch(a.batch, a.size, a.size, seed=1)
A, B, L = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev), torch.from_numpy(lab).to(dev)

def step():
    out = m(A, B)
    out = out[-1] if isinstance(out, (list, tuple)) else out
    loss = cross_entropy(out, L) if label == 2 else bce_dice_with_logits(out, L.float().unsqueeze(1))
    loss.backward()

for _ in range(3):
    step()
e = m._engine
e.profile_enable(True)
N = 3
for _ in range(N):
    step()
torch.cuda.synchronize()
k = e.profile_kernels()
e.profile_enable(False)
tot = sum(v["ms"] for v in k.values()) / N
print(f"{a.model}: instrumented {tot:.3f} ms/step")
print(f"{'kernel':40s} {'n/step':>6s} {'ms/step':>8s} {'us/launch':>9s} {'TFLOP/s':>8s} {'GB/s':>8s}")
for name, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"]):
    ms = v["ms"] / N
    print(f"{name[:40]:40s} {v['launches'] // N:6d} {ms:8.3f} {1e3 * v['ms'] / max(v['launches'], 1):9.1f} {v['flops'] / N / ms / 1e9:8.1f} {v['bytes'] / N / ms / 1e6:8.0f}")
