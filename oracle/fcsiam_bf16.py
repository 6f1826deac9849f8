"""TEST INFRASTRUCTURE ONLY -- the FC-Siam oracle (oracle/fcsiam_ref.py, pinned to the reference by tests/golden) with bf16 rounding
at exactly the points where the HIP engine's bf16 mode STORES a tensor, forward and backward, fp32 arithmetic everywhere else:

  forward   input image; filter images of every convolution (forward and data-gradient products use the bf16 filter, the weight
            gradient itself stays fp32); every conv / transposed-conv output Y (BatchNorm statistics are those of the rounded
            values, stcd_amd/csrc/kernels_conv_mfma.hip epilogues); every activation A = dropout(relu(BN(Y))); pooled maps and the
            skip fusion |a1 - a2| / a2 - a1 (formed from the rounded activations and rounded once more: k_bn_act_pair)
  backward  the gradient of every one of those tensors is rounded once where the engine writes its gradient buffer (dA: the
            data-gradient conv's output, or pool + fusion gradient summed then rounded -- k_skip_bwd; dY: k_bn_bwd_apply's output);
            d(logits) is rounded when it is packed for the last layer's gradient launches (k_gout_pack)

What this file is for: rounding noise of ~20 bf16-stored layers moves whole-network gradients far from
the fp32 reference's, so a bound against the fp32 reference alone cannot tell rounding from a wrong term.  Against THIS emulation
the engine's bf16 gradients must agree closely (tests/test_bf16_emulation_gpu.py): what is left is accumulation order and the
rare activation that sits within one fp32 ulp of a bf16 rounding boundary.
"""
from __future__ import annotations

import contextlib

import torch

from . import fcsiam_ref as R


class _Q(torch.autograd.Function):
    """round to bf16 in the forward (fwd=True) and / or round the gradient in the backward (bwd=True)"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.bfloat16().float() if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.bwd else g), None, None


def q(x, fwd=True, bwd=True):
    return _Q.apply(x, fwd, bwd)


def ste(w):
    """bf16 filter image in the products, fp32 weight gradient (straight through)"""
    return w + (w.detach().bfloat16().float() - w.detach())


@contextlib.contextmanager
def emulate():
    """Inside: oracle.fcsiam_ref.forward computes what the engine's bf16 mode stores."""
    orig = (R.conv3x3, R.convT3x3_s1, R.convT3x3_s2, R.maxpool2, R._cbrd, torch.abs)
    c3, t1, t2, mp = orig[0], orig[1], orig[2], orig[3]

    def cbrd(x, st, sfx, training, mask, conv):
        y = q(conv(x, ste(st[f"conv{sfx}.weight"]), st[f"conv{sfx}.bias"]))
        z = R.batchnorm(y, st[f"bn{sfx}.weight"], st[f"bn{sfx}.bias"], st[f"bn{sfx}.running_mean"], st[f"bn{sfx}.running_var"], training)
        if training:
            st[f"bn{sfx}.num_batches_tracked"] += 1
        a = torch.relu(z)
        if training and mask is not None:
            a = a * mask[:, :, None, None]
        return q(a)

    R._cbrd = cbrd
    R.maxpool2 = lambda x: q(mp(x), fwd=False)                                   # max of bf16 values is exact; d(pooled) is stored
    R.convT3x3_s2 = lambda x, w, b: q(t2(x, ste(w), b))                          # up-conv output inside the concat buffer
    R.convT3x3_s1 = lambda x, w, b: q(t1(x, ste(w), b), fwd=False)               # conv11d: fp32 logits, bf16 d(logits)
    try:
        yield
    finally:
        R.conv3x3, R.convT3x3_s1, R.convT3x3_s2, R.maxpool2, R._cbrd = orig[:5]


def forward(arch, st, x1, x2, masks):
    """Training-mode forward of the emulation.  The skip fusion is rounded where the oracle concatenates it."""
    fuse_q = {"diff": lambda f1, f2: q(torch.abs(f1 - f2)), "sub": lambda f1, f2: q(f2 - f1)}
    with emulate():
        if arch == "conc":
            return R.forward(arch, st, q(x1, bwd=False), q(x2, bwd=False), training=True, masks=masks)
        # diff / sub: re-state the decoder loop only to place the rounding of the fused skip (R.forward computes it inline)
        B = x1.shape[0]
        skips, bott = [], None
        for t, x in enumerate((q(x1, bwd=False), q(x2, bwd=False))):
            feats = []
            for stage in R.ENCODER:
                for sfx, _, _ in stage:
                    x = R._cbrd(x, st, sfx, True, None if masks is None else masks[f"do{sfx}"][t * B:(t + 1) * B], R.conv3x3)
                feats.append(x)
                x = R.maxpool2(x)
            skips.append(feats)
            bott = x
        x = bott
        for lvl, (up, c, convs) in zip((3, 2, 1, 0), R.DECODER):
            f1, f2 = skips[0][lvl], skips[1][lvl]
            x = R.pad_to(R.convT3x3_s2(x, st[f"{up}.weight"], st[f"{up}.bias"]), f1)
            x = torch.cat((x, fuse_q[arch](f1, f2)), 1)
            for sfx, _, _ in convs:
                if sfx == "11d":
                    x = R.convT3x3_s1(x, st["conv11d.weight"], st["conv11d.bias"])
                else:
                    x = R._cbrd(x, st, sfx, True, None if masks is None else masks[f"do{sfx}"], R.convT3x3_s1)
        return x
