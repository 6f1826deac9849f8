"""CPU restatement (torch, fp32) of SNUNet-CD with ECAM.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Follows /root/reference/models/SNUNet.py:
  conv_block_nested :8-26   (conv1 -> [identity] -> bn1 -> relu -> conv2 -> bn2 -> relu(. + identity))
  up                :29-43  (ConvTranspose2d(k=2, s=2))
  ChannelAttention  :46-59  (avg & max global pool -> fc1 -> relu -> fc2, summed, sigmoid)
  SNUNet_ECAM       :63-152 (dense nested decoder; encoder A: 4 levels, encoder B: 5 levels)

Parity status: pinned by tests/golden/g2_snunet_*.npz.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .fcsiam_ref import batchnorm, conv3x3, maxpool2

N1 = 32
FILTERS = (N1, N1 * 2, N1 * 4, N1 * 8, N1 * 16)

# registration order of SNUNet_ECAM.__init__ (SNUNet.py:73-106); ("block", name, in, mid/out) | ("up", name, ch)
def _layout(in_ch):
    f = FILTERS
    return (
        ("block", "conv0_0", in_ch, f[0]),
        ("block", "conv1_0", f[0], f[1]), ("up", "Up1_0", f[1]),
        ("block", "conv2_0", f[1], f[2]), ("up", "Up2_0", f[2]),
        ("block", "conv3_0", f[2], f[3]), ("up", "Up3_0", f[3]),
        ("block", "conv4_0", f[3], f[4]), ("up", "Up4_0", f[4]),
        ("block", "conv0_1", f[0] * 2 + f[1], f[0]),
        ("block", "conv1_1", f[1] * 2 + f[2], f[1]), ("up", "Up1_1", f[1]),
        ("block", "conv2_1", f[2] * 2 + f[3], f[2]), ("up", "Up2_1", f[2]),
        ("block", "conv3_1", f[3] * 2 + f[4], f[3]), ("up", "Up3_1", f[3]),
        ("block", "conv0_2", f[0] * 3 + f[1], f[0]),
        ("block", "conv1_2", f[1] * 3 + f[2], f[1]), ("up", "Up1_2", f[1]),
        ("block", "conv2_2", f[2] * 3 + f[3], f[2]), ("up", "Up2_2", f[2]),
        ("block", "conv0_3", f[0] * 4 + f[1], f[0]),
        ("block", "conv1_3", f[1] * 4 + f[2], f[1]), ("up", "Up1_3", f[1]),
        ("block", "conv0_4", f[0] * 5 + f[1], f[0]),
    )


def param_specs(in_ch: int, out_ch: int):
    specs = []
    for item in _layout(in_ch):
        if item[0] == "block":
            _, name, ci, co = item
            for k, (a, b) in (("1", (ci, co)), ("2", (co, co))):
                specs.append((f"{name}.conv{k}.weight", (b, a, 3, 3), "conv_w"))
                specs.append((f"{name}.conv{k}.bias", (b,), "conv_b"))
                specs.append((f"{name}.bn{k}.weight", (b,), "bn_w"))
                specs.append((f"{name}.bn{k}.bias", (b,), "bn_b"))
                specs.append((f"{name}.bn{k}.running_mean", (b,), "bn_rm"))
                specs.append((f"{name}.bn{k}.running_var", (b,), "bn_rv"))
                specs.append((f"{name}.bn{k}.num_batches_tracked", (), "bn_nbt"))
        else:
            _, name, c = item
            specs.append((f"{name}.up.weight", (c, c, 2, 2), "convT2_w"))
            specs.append((f"{name}.up.bias", (c,), "conv_b"))
    c4 = FILTERS[0] * 4
    specs.append(("ca.fc1.weight", (c4 // 16, c4, 1, 1), "fc_w"))
    specs.append(("ca.fc2.weight", (c4, c4 // 16, 1, 1), "fc_w"))
    c1 = FILTERS[0]
    specs.append(("ca1.fc1.weight", (c1 // 4, c1, 1, 1), "fc_w"))
    specs.append(("ca1.fc2.weight", (c1, c1 // 4, 1, 1), "fc_w"))
    specs.append(("conv_final.weight", (out_ch, c4, 1, 1), "fc_w"))
    specs.append(("conv_final.bias", (out_ch,), "conv_b"))
    return specs


def synth_state(in_ch: int, out_ch: int, seed: int, perturb_running: bool = False):
    import numpy as np

    rng = np.random.default_rng(seed)
    st = OrderedDict()
    for name, shape, kind in param_specs(in_ch, out_ch):
        if kind == "conv_w":
            v = rng.standard_normal(shape) * math.sqrt(2.0 / (shape[1] * 9))
        elif kind == "convT2_w":
            v = rng.standard_normal(shape) * math.sqrt(1.0 / shape[0])
        elif kind == "fc_w":
            v = rng.standard_normal(shape) * math.sqrt(2.0 / shape[1])
        elif kind == "conv_b":
            v = rng.standard_normal(shape) * 0.05
        elif kind == "bn_w":
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind == "bn_b":
            v = 0.1 * rng.standard_normal(shape)
        elif kind == "bn_rm":
            v = 0.1 * rng.standard_normal(shape) if perturb_running else np.zeros(shape)
        elif kind == "bn_rv":
            v = 1.0 + 0.3 * rng.random(shape) if perturb_running else np.ones(shape)
        elif kind == "bn_nbt":
            st[name] = torch.zeros((), dtype=torch.int64)
            continue
        st[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).clone()
    return st


def convT2x2_s2(x, w, b):
    """nn.ConvTranspose2d(C, C, 2, stride=2) (SNUNet.py:38):
    out[2i+ky, 2j+kx, co] = sum_ci in[i,j,ci] * W[ci,co,ky,kx] + b[co]."""
    n, c, h, wd = x.shape
    co = w.shape[1]
    y = torch.einsum("nchw,cokl->nohkwl", x, w).reshape(n, co, 2 * h, 2 * wd)
    return y + b[None, :, None, None]


def block(x, st, name, training):
    """conv_block_nested.forward (SNUNet.py:17-26)."""
    y1 = conv3x3(x, st[f"{name}.conv1.weight"], st[f"{name}.conv1.bias"])
    identity = y1
    z1 = batchnorm(y1, st[f"{name}.bn1.weight"], st[f"{name}.bn1.bias"],
                   st[f"{name}.bn1.running_mean"], st[f"{name}.bn1.running_var"], training)
    a1 = torch.relu(z1)
    y2 = conv3x3(a1, st[f"{name}.conv2.weight"], st[f"{name}.conv2.bias"])
    z2 = batchnorm(y2, st[f"{name}.bn2.weight"], st[f"{name}.bn2.bias"],
                   st[f"{name}.bn2.running_mean"], st[f"{name}.bn2.running_var"], training)
    if training:
        st[f"{name}.bn1.num_batches_tracked"] += 1
        st[f"{name}.bn2.num_batches_tracked"] += 1
    return torch.relu(z2 + identity)


def channel_attention(x, w1, w2):
    """ChannelAttention.forward (SNUNet.py:55-59); 1x1 convs without bias."""
    avg = x.mean(dim=(2, 3))
    mx = x.amax(dim=(2, 3))
    w1m, w2m = w1[:, :, 0, 0], w2[:, :, 0, 0]

    def mlp(v):
        return torch.relu(v @ w1m.t()) @ w2m.t()

    return torch.sigmoid(mlp(avg) + mlp(mx))[:, :, None, None]


def forward(st, xa, xb, training=False):
    def up(name, x):
        return convT2x2_s2(x, st[f"{name}.up.weight"], st[f"{name}.up.bias"])

    def blk(name, x):
        return block(x, st, name, training)

    x0_0A = blk("conv0_0", xa)
    x1_0A = blk("conv1_0", maxpool2(x0_0A))
    x2_0A = blk("conv2_0", maxpool2(x1_0A))
    x3_0A = blk("conv3_0", maxpool2(x2_0A))
    x0_0B = blk("conv0_0", xb)
    x1_0B = blk("conv1_0", maxpool2(x0_0B))
    x2_0B = blk("conv2_0", maxpool2(x1_0B))
    x3_0B = blk("conv3_0", maxpool2(x2_0B))
    x4_0B = blk("conv4_0", maxpool2(x3_0B))

    cat = lambda *t: torch.cat(t, 1)
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

    out = cat(x0_1, x0_2, x0_3, x0_4)
    intra = x0_1 + x0_2 + x0_3 + x0_4
    ca1 = channel_attention(intra, st["ca1.fc1.weight"], st["ca1.fc2.weight"])
    ca = channel_attention(out, st["ca.fc1.weight"], st["ca.fc2.weight"])
    out = ca * (out + ca1.repeat(1, 4, 1, 1))
    return F.conv2d(out, st["conv_final.weight"], st["conv_final.bias"])
