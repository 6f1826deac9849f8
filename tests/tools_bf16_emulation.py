#!/usr/bin/env python3
"""CPU experiment: how much do bf16-STORED activations / gradients (fp32 arithmetic) move the gradients of the oracle?
Rounds every tensor the engine stores in bf16 (conv outputs, activations, pooled maps, skip fusions and their gradients)
inside the oracle's forward, then compares per-tensor gradients with the plain fp32 run (G7 inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import fcsiam_ref as R

MODE = {"fwd": True, "bwd": True}


class Q(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float() if MODE["fwd"] else x

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float() if MODE["bwd"] else g


def run(arch, quant, seed=700):
    rng = np.random.default_rng(seed + 1)
    a = rng.standard_normal((2, 3, 128, 128)).astype(np.float32)
    b = (a + 0.5 * rng.standard_normal((2, 3, 128, 128))).astype(np.float32)
    tgt = torch.from_numpy((np.random.default_rng(seed + 4).random((2, 128, 128)) < 0.2).astype(np.int64))
    st = R.synth_state(arch, 3, 2, seed)
    params = [k for k, v in st.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        st[k].requires_grad_(True)
    orig = (R.conv3x3, R.convT3x3_s1, R.convT3x3_s2, R.maxpool2, torch.relu)
    if quant:
        R.conv3x3 = lambda x, w, bb: Q.apply(orig[0](x, w, bb))
        R.convT3x3_s1 = lambda x, w, bb: Q.apply(orig[1](x, w, bb))
        R.convT3x3_s2 = lambda x, w, bb: Q.apply(orig[2](x, w, bb))
        R.maxpool2 = lambda x: orig[3](Q.apply(x))          # the activation feeding pool / skip is stored in bf16
    try:
        logits = R.forward(arch, st, torch.from_numpy(a), torch.from_numpy(b), training=True, masks=R.synth_masks(arch, 2, seed + 3))
    finally:
        R.conv3x3, R.convT3x3_s1, R.convT3x3_s2, R.maxpool2 = orig[:4]
    loss = R.cross_entropy(logits, tgt)
    loss.backward()
    return loss.item(), {k: st[k].grad.clone() for k in params}


for arch in ("diff", "conc"):
    l0, g0 = run(arch, False)
    for fwd, bwd in ((True, True), (True, False), (False, True)):
        MODE["fwd"], MODE["bwd"] = fwd, bwd
        l1, g1 = run(arch, True)
        cos = {k: float((g0[k].flatten() @ g1[k].flatten()) / (g0[k].norm() * g1[k].norm() + 1e-30)) for k in g0 if g0[k].abs().max() > 1e-6 and g0[k].numel() >= 64}
        ks = sorted(cos, key=cos.get)
        print(arch, f"fwd-round={fwd} bwd-round={bwd}: loss {l0:.5f} -> {l1:.5f}; median cos {np.median(list(cos.values())):.4f}; worst", [(k, round(cos[k], 3)) for k in ks[:4]])
