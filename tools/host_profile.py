#!/usr/bin/env python3
"""Where the HOST's time of a training step goes (cProfile over N steps with the GPU idle between steps):
    python3 tools/host_profile.py [changeformer|mitb0|diff|snunet|segcd] [N]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stcd_amd import synth
from stcd_amd.optim import FlatAdamW

which = sys.argv[1] if len(sys.argv) > 1 else "changeformer"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
if which in ("changeformer", "mitb0"):
    from stcd_amd.changeformer import ChangeFormerV6, MIT_B0
    m = ChangeFormerV6(3, 2, dtype="bf16", config=dict(MIT_B0) if which == "mitb0" else None).to(dev).train()
    B, S = 4, 512
elif which == "snunet":
    from stcd_amd.modules import SNUNet_ECAM
    m = SNUNet_ECAM(3, 2, dtype="bf16").to(dev).train(); B, S = 16, 256
elif which == "segcd":
    from stcd_amd.segcd import SegCD
    m = SegCD(encoder_name="resnet50", dtype="bf16").to(dev).train(); B, S = 16, 256
else:
    from stcd_amd.modules import SiamUnet_diff
    m = SiamUnet_diff(3, 2, dtype="bf16").to(dev).train(); B, S = 16, 256
opt = FlatAdamW(m, lr=1e-3)
a, b, lab = synth.make_batch(B, S, S, seed=1)
A, Bt, L = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(lab).to(dev)
from stcd_amd.losses import cross_entropy, bce_dice_with_logits


def step():
    opt.zero_grad(set_to_none=True)
    out = m(A, Bt)
    out = out[-1] if isinstance(out, (list, tuple)) else out
    loss = cross_entropy(out, L) if out.shape[1] == 2 else bce_dice_with_logits(out, L.float().unsqueeze(1))
    loss.backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
ts = []
pr = cProfile.Profile()
for _ in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pr.enable(); step(); pr.disable(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print(f"{which}: host enqueue (GPU idle) median {1e3 * sorted(ts)[N // 2]:.3f} ms/step")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
