"""TEST INFRASTRUCTURE ONLY -- the SNUNet-ECAM oracle (oracle/snunet_ref.py, pinned to the reference by tests/golden/g2_snunet_* /
g7_snunet_*) with bf16 rounding at exactly the points where the HIP engine's bf16 mode STORES a tensor, forward and backward, fp32
arithmetic everywhere else (the construction of oracle/fcsiam_bf16.py; reference: /root/reference/models/SNUNet.py:8-26,116-152).

  forward   the input images; the filter image of every convolution / transposed convolution (products use the bf16 filter, the
            weight gradient stays fp32: straight-through); conv1's raw output Y1 (which is ALSO the identity branch, SNUNet.py:19);
            A1 = relu(bn1(Y1)); Y2; the block output relu(bn2(Y2) + Y1) -- written once per consumer concat slice with the same
            bits (k_bn_act's extra destinations); the 2x2 transposed convs' outputs inside the consumers' concat buffers; the ECAM
            output Z = ca * (E + ca1) (the 1x1 head reads it).  BatchNorm statistics are those of the ROUNDED conv outputs.
  backward  d(logits) where k_gout_pack packs it; dZ (the head's data gradient); dE (k_ecam_bwd2's output); d(concat input) of
            every block (conv1's data gradient: one rounding for all slices) and the transposed convs' data gradients -- each is ONE
            CONTRIBUTION to its producer's output gradient; the producer's reduction (k_bn_reduce<.., NS>) sums the contributions in
            fp32 and rounds ONCE more; dZ2 (the gated gradient, which is also the identity branch's gradient into Y1: stored, then
            added to bn1's backward in fp32); dY2, dA1, dY1 (k_bn_bwd_apply / the data-gradient conv outputs).

Against THIS emulation the engine's bf16 gradients must agree closely (tests/test_bf16_emulation_gpu.py): what is left is
accumulation order and activations within one fp32 ulp of a rounding boundary.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import snunet_ref as S
from .fcsiam_bf16 import q, ste
from .fcsiam_ref import batchnorm, conv3x3, maxpool2


def _block(x, st, name):
    """conv_block_nested (SNUNet.py:17-26), training mode, with the engine's bf16 stores (sn_block_forward / sn_block_backward)."""
    y1 = q(conv3x3(x, ste(st[f"{name}.conv1.weight"]), st[f"{name}.conv1.bias"]))          # Y1 stored; dY1 stored
    identity = q(y1, fwd=False)                                                               # dZ2: stored, then added in fp32
    z1 = batchnorm(y1, st[f"{name}.bn1.weight"], st[f"{name}.bn1.bias"], st[f"{name}.bn1.running_mean"],
                   st[f"{name}.bn1.running_var"], True)
    a1 = q(torch.relu(z1))                                                                    # A1 stored; dA1 stored
    y2 = q(conv3x3(a1, ste(st[f"{name}.conv2.weight"]), st[f"{name}.conv2.bias"]))          # Y2 stored; dY2 stored
    z2 = batchnorm(y2, st[f"{name}.bn2.weight"], st[f"{name}.bn2.bias"], st[f"{name}.bn2.running_mean"],
                   st[f"{name}.bn2.running_var"], True)
    st[f"{name}.bn1.num_batches_tracked"] += 1
    st[f"{name}.bn2.num_batches_tracked"] += 1
    return q(torch.relu(z2 + identity))                                                      # Out stored; its summed gradient rounded once


def forward(st, xa, xb):
    """Training-mode forward of the emulation (SNUNet_ECAM.forward, SNUNet.py:116-152)."""
    def up(name, x):        # the up-conv's data gradient is one (rounded) contribution to x's gradient; its output is stored
        return q(S.convT2x2_s2(q(x, fwd=False), ste(st[f"{name}.up.weight"]), st[f"{name}.up.bias"]))

    def blk(name, x):
        return _block(x, st, name)

    pool = lambda x: q(maxpool2(x), fwd=False)                 # max of bf16 values is exact; d(pooled) = the consumer's dIn (stored)
    cat = lambda *t: q(torch.cat(t, 1), fwd=False)             # d(concat input): conv1's data gradient, one rounding for all slices
    xa, xb = q(xa, bwd=False), q(xb, bwd=False)
    x0_0A = blk("conv0_0", xa)
    x1_0A = blk("conv1_0", pool(x0_0A))
    x2_0A = blk("conv2_0", pool(x1_0A))
    x3_0A = blk("conv3_0", pool(x2_0A))
    x0_0B = blk("conv0_0", xb)
    x1_0B = blk("conv1_0", pool(x0_0B))
    x2_0B = blk("conv2_0", pool(x1_0B))
    x3_0B = blk("conv3_0", pool(x2_0B))
    x4_0B = blk("conv4_0", pool(x3_0B))

    x0_1 = blk("conv0_1", cat(x0_0A, x0_0B, up("Up1_0", x1_0B)))
    x1_1 = blk("conv1_1", cat(x1_0A, x1_0B, up("Up2_0", x2_0B)))
    x0_2 = blk("conv0_2", cat(x0_0A, x0_0B, x0_1, up("Up1_1", x1_1)))
    x2_1 = blk("conv2_1", cat(x2_0A, x2_0B, up("Up3_0", x3_0B)))
    x1_2 = blk("conv1_2", cat(x1_0A, x1_0B, x1_1, up("Up2_1", x2_1)))
    x0_3 = blk("conv0_3", cat(x0_0A, x0_0B, x0_1, x0_2, up("Up1_2", x1_2)))
    x3_1 = blk("conv3_1", cat(x3_0A, x3_0B, up("Up4_0", x4_0B)))
    x2_2 = blk("conv2_2", cat(x2_0A, x2_0B, x2_1, up("Up3_1", x3_1)))
    x1_3 = blk("conv1_3", cat(x1_0A, x1_0B, x1_1, x1_2, up("Up2_2", x2_2)))
    x0_4 = blk("conv0_4", cat(x0_0A, x0_0B, x0_1, x0_2, x0_3, up("Up1_3", x1_3)))

    out = cat(x0_1, x0_2, x0_3, x0_4)                          # E (the blocks' own stores); dE: one rounding (k_ecam_bwd2)
    intra = out[:, :S.N1] + out[:, S.N1:2 * S.N1] + out[:, 2 * S.N1:3 * S.N1] + out[:, 3 * S.N1:]
    ca1 = S.channel_attention(intra, st["ca1.fc1.weight"], st["ca1.fc2.weight"])
    ca = S.channel_attention(out, st["ca.fc1.weight"], st["ca.fc2.weight"])
    z = q(ca * (out + ca1.repeat(1, 4, 1, 1)))                 # Z stored; dZ stored
    logits = F.conv2d(z, ste(st["conv_final.weight"]), st["conv_final.bias"])
    return q(logits, fwd=False)                                # fp32 logits; d(logits) rounded when packed
