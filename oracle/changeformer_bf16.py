"""TEST INFRASTRUCTURE ONLY -- the ChangeFormerV6 oracle (oracle/changeformer_ref.py: decoder pinned to the reference's own classes by
tests/golden/g21_cf_decoder.npz, encoder restated from the text -- parity unpinned, see that file's header) with bf16 rounding at
exactly the tensors the HIP engine's bf16 mode STORES (stcd_amd/csrc/engine_cf.inl: every CfT has a value and a gradient buffer),
forward and backward, fp32 arithmetic everywhere else.  The construction of oracle/fcsiam_bf16.py / snunet_bf16.py.

  forward   the input images; every GEMM-shaped filter (products use the bf16 image, the weight gradient stays fp32: straight-through;
            LayerNorm / BatchNorm / PReLU / depth-wise parameters are read in fp32 by their kernels); per encoder stage: the patch
            embedding's conv output (peo) and its LayerNorm (tok), and per block xn = LN1(x), q, [st = sr-conv(xn), sn = LN(st)], kv,
            the attention output ao (softmax probabilities enter the P.V product as bf16: k_attn_fwd's second MFMA), pr = proj(ao),
            x1 = x + drop_path(drop(pr)), xn2 = LN2(x1), hd = fc1, u = depth-wise conv + bias (the pre-activation is kept),
            a = drop(gelu(u)), f2 = fc2(a), x2; the stage's output LayerNorm.  Decoder: lo = linear_c (both dates), ya / za = prelu /
            ba = bn / aa = dropout and the same for the second conv of conv_diff, c (+ the up-sampled coarser c: a second store of the
            same buffer), the bilinear slices of the fusion concat, fy / fa, the transposed convs' outputs, and in every ResidualBlock
            r1 = relu(conv1) and out = 0.1 * round(conv2) + x (ConvEpi: the conv result is rounded before the scale-add).  The four
            auxiliary heads and cp are fp32.
  backward  the gradient of every one of those tensors, rounded where the engine writes its gradient buffer; contributions that are
            stored separately and summed later (q / kv data gradients into xn, the two halves of the transposed convs' data gradient,
            a coarser scale's c feeding both the next scale and the fusion concat) are rounded per contribution.

Against THIS emulation the engine's bf16 gradients must agree closely (tests/test_bf16_emulation_gpu.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import changeformer_ref as R
from .fcsiam_bf16 import q, ste

qb = lambda x: q(x, fwd=False, bwd=True)       # a separately stored gradient contribution


def _lin(st, name, x):
    return q(F.linear(x, ste(st[name + ".weight"]), st[name + ".bias"]))


def _ln(st, name, x, eps):
    return q(F.layer_norm(x, (x.shape[-1],), st[name + ".weight"], st[name + ".bias"], eps))


def _attention(st, name, xn, H, W, heads, sr, masks):
    B, N, C = xn.shape
    d = C // heads
    qq = _lin(st, name + ".q", qb(xn)).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = qb(xn).permute(0, 2, 1).reshape(B, C, H, W)            # (col2im accumulates onto the q path's stored gradient)
        x_ = q(F.conv2d(x_, ste(st[name + ".sr.weight"]), st[name + ".sr.bias"], stride=sr)).reshape(B, C, -1).permute(0, 2, 1)
        x_ = _ln(st, name + ".norm", x_, 1e-5)
    else:
        x_ = qb(xn)
    kv = _lin(st, name + ".kv", x_).reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = ((qq @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    attn = q(R._m(masks, name + ".attn_drop", attn), bwd=False)      # P as the bf16 operand of the second MFMA
    x = q((attn @ v).transpose(1, 2).reshape(B, N, C))               # ao
    return _lin(st, name + ".proj", x)                               # pr (proj_drop is applied inside k_resid_drop)


def _block(st, name, x, H, W, heads, sr, masks):
    xn = _ln(st, name + ".norm1", x, 1e-6)
    pr = _attention(st, name + ".attn", xn, H, W, heads, sr, masks)
    x1 = q(x + R._m(masks, name + ".drop_path1", R._m(masks, name + ".attn.proj_drop", pr)))
    xn2 = _ln(st, name + ".norm2", x1, 1e-6)
    hd = _lin(st, name + ".mlp.fc1", xn2)
    u = q(R.dwconv_tokens(hd, st[name + ".mlp.dwconv.dwconv.weight"], st[name + ".mlp.dwconv.dwconv.bias"], H, W))
    a = q(R._m(masks, name + ".mlp.drop1", F.gelu(u)))
    f2 = _lin(st, name + ".mlp.fc2", a)
    return q(x1 + R._m(masks, name + ".drop_path2", R._m(masks, name + ".mlp.drop2", f2)))


def _encoder(cfg, st, x, masks):
    outs = []
    B = x.shape[0]
    for s in range(4):
        k, stride = (cfg.patch1, 4) if s == 0 else (cfg.patch, 2)
        name = f"Tenc_x2.patch_embed{s + 1}"
        y = q(F.conv2d(x, ste(st[name + ".proj.weight"]), st[name + ".proj.bias"], stride=stride, padding=k // 2))
        _, C, H, W = y.shape
        t = _ln(st, name + ".norm", y.flatten(2).transpose(1, 2), 1e-5)
        for i in range(cfg.depths[s]):
            t = _block(st, f"Tenc_x2.block{s + 1}.{i}", t, H, W, cfg.num_heads[s], cfg.sr_ratios[s], masks)
        t = _ln(st, f"Tenc_x2.norm{s + 1}", t, 1e-6)
        x = t.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


def _conv_diff(st, name, x, masks):
    x = q(F.conv2d(x, ste(st[name + ".0.weight"]), st[name + ".0.bias"], padding=1))          # ya
    x = q(F.prelu(x, st[name + ".1.weight"]))                                                    # za
    x = q(R._bn(st, name + ".2", x, True))                                                       # ba
    x = q(R._m(masks, name + ".3", x))                                                           # aa
    x = q(F.conv2d(x, ste(st[name + ".4.weight"]), st[name + ".4.bias"], padding=1))          # yb
    x = q(F.prelu(x, st[name + ".5.weight"]))                                                    # zb
    x = q(R._bn(st, name + ".6", x, True))                                                       # bb
    return q(R._m(masks, name + ".7", x))                                                        # c (before the coarser scale is added)


def _res(st, name, x):
    r1 = q(F.relu(R.conv_layer(x, ste(st[name + ".conv1.conv2d.weight"]), st[name + ".conv1.conv2d.bias"], 1)))
    y2 = q(R.conv_layer(r1, ste(st[name + ".conv2.conv2d.weight"]), st[name + ".conv2.conv2d.bias"], 1))      # rounded before the scale-add
    return q(y2 * 0.1 + x)


def forward(cfg: R.CFConfig, st, x1, x2, masks):
    """Training-mode forward of the emulation -> [p_c4, p_c3, p_c2, p_c1, cp] (ChangeFormerV6.forward, ChangeFormer.py:1693-1701)."""
    B = x1.shape[0]
    m1 = {k: (v[:B] if k.startswith("Tenc_x2.") else v) for k, v in masks.items()} if masks is not None else None
    m2 = {k: (v[B:] if k.startswith("Tenc_x2.") else v) for k, v in masks.items()} if masks is not None else None
    f1 = _encoder(cfg, st, q(x1, bwd=False), m1)
    f2 = _encoder(cfg, st, q(x2, bwd=False), m2)
    d = "TDec_x2."
    size1 = f1[0].shape[2:]
    outs, prev, ups = [], None, []
    for s in (4, 3, 2, 1):
        a, b = f1[s - 1], f2[s - 1]
        n, _, h, w = a.shape

        def mlp(x):
            y = _lin(st, d + f"linear_c{s}.proj", x.flatten(2).transpose(1, 2))
            return y.permute(0, 2, 1).reshape(n, -1, h, w)

        c = _conv_diff(st, d + f"diff_c{s}", torch.cat((mlp(a), mlp(b)), dim=1), masks)
        if prev is not None:
            c = q(c + F.interpolate(qb(prev), scale_factor=2, mode="bilinear"))
        x_aux = F.relu(F.conv2d(qb(c), ste(st[d + f"make_pred_c{s}.0.weight"]), st[d + f"make_pred_c{s}.0.bias"], padding=1))
        x_aux = R._bn(st, d + f"make_pred_c{s}.2", x_aux, True)
        outs.append(F.conv2d(x_aux, st[d + f"make_pred_c{s}.3.weight"], st[d + f"make_pred_c{s}.3.bias"], padding=1))
        ups.append(c if s == 1 else q(F.interpolate(qb(c), size=size1, mode="bilinear", align_corners=False)))
        prev = c
    x = q(F.conv2d(torch.cat(ups, dim=1), ste(st[d + "linear_fuse.0.weight"]), st[d + "linear_fuse.0.bias"]))
    x = q(R._bn(st, d + "linear_fuse.1", x, True))
    x = q(R.upsample_conv(x, ste(st[d + "convd2x.conv2d.weight"]), st[d + "convd2x.conv2d.bias"]))
    x = _res(st, d + "dense_2x.0", x)
    x = q(R.upsample_conv(x, ste(st[d + "convd1x.conv2d.weight"]), st[d + "convd1x.conv2d.bias"]))
    x = _res(st, d + "dense_1x.0", x)
    cp = R.conv_layer(x, ste(st[d + "change_probability.conv2d.weight"]), st[d + "change_probability.conv2d.bias"], 1)
    outs.append(q(cp, fwd=False))                    # fp32 logits; d(logits) is rounded when it is packed
    return outs
