"""Stability soak: N training steps of one family in bf16 on a small fixed synthetic set; prints the loss every 10 % and checks that
everything stays finite.  python tools/soak.py --model segcd --steps 2000"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stcd_amd import synth
from stcd_amd.losses import bce_dice_with_logits, cross_entropy
from stcd_amd.optim import FlatAdamW
ap = argparse.ArgumentParser(); ap.add_argument("--model", default="diff"); ap.add_argument("--steps", type=int, default=1000)
a = ap.parse_args()
dev = "cuda:0"
if a.model == "segcd":
    from stcd_amd.segcd import SegCD
    m, label = SegCD().to(dev).train(), 1
else:
    from stcd_amd import modules
    m = {"diff": modules.SiamUnet_diff, "conc": modules.SiamUnet_conc, "snunet": modules.SNUNet_ECAM}[a.model](3, 2).to(dev).train(); label = 2
opt = FlatAdamW(m, lr=1e-3, weight_decay=0.01)
x1, x2, lab = synth.make_batch(64, 128, 128, seed=5)
A, B, L = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev), torch.from_numpy(lab).to(dev)
losses = []
for s in range(a.steps):
    i = (s * 16) % 64
    opt.zero_grad(set_to_none=True)
    out = m(A[i:i + 16], B[i:i + 16]); out = out[-1] if isinstance(out, (list, tuple)) else out
    loss = cross_entropy(out, L[i:i + 16]) if label == 2 else bce_dice_with_logits(out, L[i:i + 16].float().unsqueeze(1))
    loss.backward(); opt.step()
    losses.append(loss.detach())
    if (s + 1) % max(1, a.steps // 10) == 0:
        print(s + 1, float(torch.stack(losses[-20:]).mean()), flush=True)
ls = torch.stack(losses)
assert torch.isfinite(ls).all(), "non-finite loss"
assert all(torch.isfinite(p).all() for p in m.parameters()), "non-finite parameter"
print("ok: first", float(ls[:20].mean()), "last", float(ls[-20:].mean()))
