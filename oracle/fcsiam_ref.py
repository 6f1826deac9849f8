"""CPU restatement (torch, fp32) of the FC-Siam family forward pass.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Follows, without sharing code with, the reference models:
  * SiamUnet_diff  -- /root/reference/models/SiamUnet_diff.py:13-92 (layers), :94-181 (forward)
  * SiamUnet_conc  -- /root/reference/models/SiamUnet_conc.py:54,66,78,87 (decoder widths), :149-172 (concat skips)
  * SiamUnet_sub   -- /root/reference/models/SiamUnet_sub.py:150-180 (signed skips, list return)
  * SiamUnet_cross_conc -- /root/reference/models/SiamUnet_crossconc.py:11-33 (cross_conc block), :119-122 (one per level), :180-204
                      (skips = cross_conc(f1, f2)); list return
  * Unet (FC-EF)   -- /root/reference/models/Unet.py:10-91 (layers: conv11 takes 2 * input_nbr channels), :93-154 (one stream over
                      cat(x1, x2), skips = the stream's own activations)

The network is written as a table walk with every non-conv op restated explicitly
(transposed convs as flipped/transposed direct convs, batch-norm from its defining
sums, dropout as an explicit per-(n,c) scale), so that agreement with the reference
(tests/golden) is a real check of the semantics the HIP engine implements.
Backward comes from CPU autograd over these explicit ops.

Parity status: pinned by tests/golden/g2_*.npz, g3_*.npz, g4_*.npz, g6_*.npz (FC-EF: g2_fcef_*.npz, g7_fcef_128.npz; cross_conc: g2_xconc_*.npz, g7_xconc_128.npz).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default (SiamUnet_diff.py:19)
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default
DROP_P = 0.2        # nn.Dropout2d(p=0.2) (SiamUnet_diff.py:20)

# encoder: stage -> list of (suffix, c_in, c_out); c_in None = network input channels
ENCODER = (
    (("11", None, 16), ("12", 16, 16)),
    (("21", 16, 32), ("22", 32, 32)),
    (("31", 32, 64), ("32", 64, 64), ("33", 64, 64)),
    (("41", 64, 128), ("42", 128, 128), ("43", 128, 128)),
)
# decoder: (upconv name, channels, [(suffix, c_in_after_concat_marker, c_out), ...])
# c_in == "cat" means (upsampled channels + skip channels)
DECODER = (
    ("upconv4", 128, (("43d", "cat", 128), ("42d", 128, 128), ("41d", 128, 64))),
    ("upconv3", 64, (("33d", "cat", 64), ("32d", 64, 64), ("31d", 64, 32))),
    ("upconv2", 32, (("22d", "cat", 32), ("21d", 32, 16))),
    ("upconv1", 16, (("12d", "cat", 16), ("11d", 16, None))),  # 11d: -> label_nbr, no BN
)
ARCHS = ("diff", "conc", "sub", "fcef", "xconc")


def skip_channels(arch: str, c: int) -> int:
    return 2 * c if arch == "conc" else c


def param_specs(arch: str, in_ch: int, label: int):
    """(name, shape, kind) in the reference's registration order.

    kind in {conv_w, conv_b, convT_w, convT_b, bn_w, bn_b, bn_rm, bn_rv, bn_nbt}.
    Conv2d weights are [Co,Ci,3,3]; ConvTranspose2d weights are [Ci,Co,3,3]
    (e.g. conv43d.weight = (256,128,3,3)).
    """
    specs = []

    def bn(name, c):
        specs.append((f"{name}.weight", (c,), "bn_w"))
        specs.append((f"{name}.bias", (c,), "bn_b"))
        specs.append((f"{name}.running_mean", (c,), "bn_rm"))
        specs.append((f"{name}.running_var", (c,), "bn_rv"))
        specs.append((f"{name}.num_batches_tracked", (), "bn_nbt"))

    for stage in ENCODER:
        for sfx, ci, co in stage:
            ci = (2 * in_ch if arch == "fcef" else in_ch) if ci is None else ci
            specs.append((f"conv{sfx}.weight", (co, ci, 3, 3), "conv_w"))
            specs.append((f"conv{sfx}.bias", (co,), "conv_b"))
            bn(f"bn{sfx}", co)
    for up, c, convs in DECODER:
        specs.append((f"{up}.weight", (c, c, 3, 3), "convT_w"))
        specs.append((f"{up}.bias", (c,), "convT_b"))
        for sfx, ci, co in convs:
            ci = c + skip_channels(arch, c) if ci == "cat" else ci
            co = label if co is None else co
            specs.append((f"conv{sfx}.weight", (ci, co, 3, 3), "convT_w"))
            specs.append((f"conv{sfx}.bias", (co,), "convT_b"))
            if sfx != "11d":
                bn(f"bn{sfx}", co)
    if arch == "xconc":      # cross_conc1..4 (SiamUnet_crossconc.py:119-122), registered after the decoder
        for l, c in enumerate((16, 32, 64, 128), 1):
            specs.append((f"cross_conc{l}.diff.0.weight", (c, 2, 3, 3), "conv_w"))       # Conv2d(2c, c, 3, groups=c)
            specs.append((f"cross_conc{l}.diff.0.bias", (c,), "conv_b"))
            bn(f"cross_conc{l}.diff.1", c)
            specs.append((f"cross_conc{l}.conv_res.0.weight", (c, c, 3, 3), "conv_w"))
            specs.append((f"cross_conc{l}.conv_res.0.bias", (c,), "conv_b"))
            bn(f"cross_conc{l}.conv_res.1", c)
    return specs


def dropout_layers(arch: str = "diff"):
    """Dropout2d call order of one forward: (name, channels, 'enc'|'dec')."""
    out = []
    for stage in ENCODER:
        for sfx, _, co in stage:
            out.append((f"do{sfx}", co, "enc"))
    for _, _, convs in DECODER:
        for sfx, _, co in convs:
            if sfx != "11d":
                out.append((f"do{sfx}", co, "dec"))
    return out


def synth_state(arch: str, in_ch: int, label: int, seed: int, perturb_running: bool = False):
    """Deterministic, reference-free parameter set (numpy Generator -> torch fp32).

    Used by both the golden generator (loaded into the reference model) and the
    tests (loaded into oracle / HIP engine), so fixtures need not carry weights.
    """
    import numpy as np

    rng = np.random.default_rng(seed)
    st = OrderedDict()
    for name, shape, kind in param_specs(arch, in_ch, label):
        if kind in ("conv_w", "convT_w"):
            fan = shape[1] * 9 if kind == "conv_w" else shape[0] * 9
            v = rng.standard_normal(shape) * math.sqrt(2.0 / fan)
        elif kind in ("conv_b", "convT_b"):
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


def synth_masks(arch: str, batch: int, seed: int, p: float = DROP_P):
    """Dropout2d masks as an explicit input: name -> [N_total, C] of {0, 1/(1-p)}.

    Encoder layers see T1's batch then T2's batch: rows [0,B) belong to the first
    call of the (shared) dropout module, rows [B,2B) to the second
    (SiamUnet_diff.py:99-119 then :123-143).
    """
    import numpy as np

    rng = np.random.default_rng(seed)
    masks = OrderedDict()
    for name, c, where in dropout_layers(arch):
        n = 2 * batch if (where == "enc" and arch != "fcef") else batch
        keep = (rng.random((n, c)) >= p).astype(np.float32) / (1.0 - p)
        masks[name] = torch.from_numpy(keep)
    return masks


# --------------------------------------------------------------------------- ops
def conv3x3(x, w, b):
    """nn.Conv2d(k=3, padding=1) (SiamUnet_diff.py:18)."""
    return F.conv2d(x, w, b, stride=1, padding=1)


def convT3x3_s1(x, w, b):
    """nn.ConvTranspose2d(k=3, padding=1) (SiamUnet_diff.py:54) as a direct conv.

    out[co] = sum_ci x[ci] (*) W'[co,ci], W'[co,ci,ky,kx] = W[ci,co,2-ky,2-kx].
    """
    return F.conv2d(x, w.flip(2, 3).transpose(0, 1), b, stride=1, padding=1)


def convT3x3_s2(x, w, b):
    """nn.ConvTranspose2d(k=3, padding=1, stride=2, output_padding=1) (SiamUnet_diff.py:52).

    Zero-insertion form: dilate the input by 2, pad (k-1-p)=1 top/left and
    (k-1-p+output_padding)=2 bottom/right, then a valid conv with the flipped kernel.
    out[2iy-1+ky, 2ix-1+kx] += in[iy,ix] * W[ci,co,ky,kx].
    """
    n, c, h, wd = x.shape
    xz = x.new_zeros(n, c, 2 * h - 1, 2 * wd - 1)
    xz[:, :, ::2, ::2] = x
    xz = F.pad(xz, (1, 2, 1, 2))
    return F.conv2d(xz, w.flip(2, 3).transpose(0, 1), b, stride=1, padding=0)


def batchnorm(x, gamma, beta, rmean, rvar, training, update_running=True):
    """nn.BatchNorm2d forward (train: batch stats + running update; eval: running stats)."""
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))  # biased
        if update_running:
            with torch.no_grad():
                rmean.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                rvar.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
    else:
        mean, var = rmean, rvar
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


def maxpool2(x):
    """F.max_pool2d(kernel_size=2, stride=2) (SiamUnet_diff.py:101)."""
    return F.max_pool2d(x, kernel_size=2, stride=2)


def pad_to(x, ref):
    """ReplicationPad2d((0, dw, 0, dh)) (SiamUnet_diff.py:149); identity when sizes agree."""
    dh, dw = ref.shape[2] - x.shape[2], ref.shape[3] - x.shape[3]
    if dh == 0 and dw == 0:
        return x
    return F.pad(x, (0, dw, 0, dh), mode="replicate")


def _cbrd(x, st, sfx, training, mask, conv):
    """conv -> BN -> ReLU -> Dropout2d (e.g. SiamUnet_diff.py:99)."""
    y = conv(x, st[f"conv{sfx}.weight"], st[f"conv{sfx}.bias"])
    z = batchnorm(y, st[f"bn{sfx}.weight"], st[f"bn{sfx}.bias"],
                  st[f"bn{sfx}.running_mean"], st[f"bn{sfx}.running_var"], training)
    if training:
        st[f"bn{sfx}.num_batches_tracked"] += 1
    a = torch.relu(z)
    if training and mask is not None:
        a = a * mask[:, :, None, None]
    return a


def _cross_conc(st, l, f1, f2, training):
    """cross_conc.forward (SiamUnet_crossconc.py:24-33): channels interleaved (0::2 = date 1, 1::2 = date 2), grouped 3x3 conv with one
    group per channel pair, BatchNorm, ReLU, 3x3 conv, BatchNorm, ReLU."""
    n, c, h, w = f1.shape
    pair = torch.stack((f1, f2), 2).reshape(n, 2 * c, h, w)
    p = f"cross_conc{l}."
    y = F.conv2d(pair, st[p + "diff.0.weight"], st[p + "diff.0.bias"], padding=1, groups=c)
    z = batchnorm(y, st[p + "diff.1.weight"], st[p + "diff.1.bias"], st[p + "diff.1.running_mean"], st[p + "diff.1.running_var"], training)
    if training:
        st[p + "diff.1.num_batches_tracked"] += 1
    y2 = conv3x3(torch.relu(z), st[p + "conv_res.0.weight"], st[p + "conv_res.0.bias"])
    z2 = batchnorm(y2, st[p + "conv_res.1.weight"], st[p + "conv_res.1.bias"], st[p + "conv_res.1.running_mean"],
                   st[p + "conv_res.1.running_var"], training)
    if training:
        st[p + "conv_res.1.num_batches_tracked"] += 1
    return torch.relu(z2)


def forward(arch, st, x1, x2, training=False, masks=None):
    """Returns logits [B,label,H,W] (tensor for every arch; the reference's list
    wrapping for 'sub' (SiamUnet_sub.py:177-180) is a boundary concern).
    ``st`` is mutated in training mode (running stats), exactly like the modules.
    """
    assert arch in ARCHS
    B = x1.shape[0]
    skips = []
    bott = None
    streams = (torch.cat((x1, x2), 1),) if arch == "fcef" else (x1, x2)      # Unet.py:94
    for t, x in enumerate(streams):
        feats = []
        for stage in ENCODER:
            for sfx, _, _ in stage:
                m = None if masks is None else masks[f"do{sfx}"][t * B:(t + 1) * B]
                x = _cbrd(x, st, sfx, training, m, conv3x3)
            feats.append(x)
            x = maxpool2(x)
        skips.append(feats)
        bott = x  # T2's pooled map overwrites T1's (SiamUnet_diff.py:119 then :143)
    x = bott
    for lvl, (up, c, convs) in zip((3, 2, 1, 0), DECODER):
        f1, f2 = skips[0][lvl], skips[-1][lvl]
        x = convT3x3_s2(x, st[f"{up}.weight"], st[f"{up}.bias"])
        x = pad_to(x, f1)
        if arch == "xconc":
            x = torch.cat((x, _cross_conc(st, lvl + 1, f1, f2, training)), 1)   # SiamUnet_crossconc.py:180
        elif arch == "fcef":
            x = torch.cat((x, f1), 1)                    # Unet.py:128
        elif arch == "diff":
            x = torch.cat((x, torch.abs(f1 - f2)), 1)   # SiamUnet_diff.py:150
        elif arch == "sub":
            x = torch.cat((x, f2 - f1), 1)               # SiamUnet_sub.py:150
        else:
            x = torch.cat((x, f1, f2), 1)                # SiamUnet_conc.py:149
        for sfx, _, _ in convs:
            if sfx == "11d":
                x = convT3x3_s1(x, st["conv11d.weight"], st["conv11d.bias"])
            else:
                m = None if masks is None else masks[f"do{sfx}"]
                x = _cbrd(x, st, sfx, training, m, convT3x3_s1)
    return x


# ------------------------------------------------------------------------ losses
def cross_entropy(logits, target, ignore_index=255):
    """/root/reference/models/losses.py:6-21 for same-size inputs: mean NLL of
    log-softmax over non-ignored pixels. target: [B,H,W] or [B,1,H,W], any int/float dtype."""
    t = target.long()
    if t.dim() == 4:
        t = t[:, 0]
    lse = torch.logsumexp(logits, dim=1)
    valid = t != ignore_index
    tc = torch.where(valid, t, torch.zeros_like(t))
    picked = torch.gather(logits, 1, tc[:, None])[:, 0]
    nll = (lse - picked) * valid
    return nll.sum() / valid.sum()


class _BCEMean(torch.autograd.Function):
    """torch.nn.BCELoss(reduction='mean') semantics, both directions:
    forward clamps each log term at -100; backward is (p - t) / max(p (1 - p), 1e-12) / N
    (ATen binary_cross_entropy / binary_cross_entropy_backward)."""

    @staticmethod
    def forward(ctx, prob, target):
        ctx.save_for_backward(prob, target)
        logp = torch.clamp(torch.log(prob), min=-100.0)
        log1mp = torch.clamp(torch.log(1.0 - prob), min=-100.0)
        return -(target * logp + (1.0 - target) * log1mp).mean()

    @staticmethod
    def backward(ctx, g):
        prob, target = ctx.saved_tensors
        return g * (prob - target) / torch.clamp(prob * (1.0 - prob), min=1e-12) / prob.numel(), None


def cd_loss(prob, target):
    """/root/reference/models/losses.py:24-34 == BCE_DICE (train_pse_cd.py:436-462):
    BCELoss(mean) with log clamped at -100, plus Dice with smooth=1."""
    bce = _BCEMean.apply(prob, target)
    inter = (prob * target).sum()
    dice = 1.0 - (2.0 * inter + 1.0) / (prob.sum() + target.sum() + 1.0)
    return dice + bce


def contrastive_loss(pred, cd_label, pse_label):
    """/root/reference/train_stcd.py:334-385: pred = cat(cd_pred, pse_pred) on dim 0 (probabilities);
    M = (cd_label == pse_label), N = its complement;
    sum((pse - cd)^2 M) / (sum M + 1e-8) + sum((pse - |cd - 1|)^2 N) / (sum N + 1e-8)."""
    b = cd_label.shape[0]
    cd_pred, pse_pred = pred[:b], pred[b:]
    M = (cd_label == pse_label).to(pred.dtype)
    N = 1.0 - M
    pos = ((pse_pred - cd_pred) ** 2 * M).sum() / (M.sum() + 1e-8)
    neg = ((pse_pred - torch.abs(cd_pred - 1.0)) ** 2 * N).sum() / (N.sum() + 1e-8)
    return pos + neg


# ----------------------------------------------------------------------- metrics
def confusion_matrix(pred, label, num_class=2):
    """train_pse_cd.py:361-368: bincount(numClass*label + pred) -> [label, pred] counts (float64)."""
    idx = num_class * label.flatten().long() + pred.flatten().long()
    return torch.bincount(idx, minlength=num_class ** 2).reshape(num_class, num_class).double()


def scores_from_cm(cm):
    """train_pse_cd.py:313-350: per-class precision/recall/F1/IoU and OA."""
    diag = torch.diag(cm)
    prec = diag / cm.sum(0)
    rec = diag / cm.sum(1)
    f1 = 2 * prec * rec / (prec + rec)
    iou = diag / (cm.sum(1) + cm.sum(0) - diag)
    oa = diag.sum() / cm.sum()
    return {"precision": prec, "recall": rec, "f1": f1, "iou": iou, "oa": oa}


def poly_lr(base_lr, epoch0, it_in_epoch, iters_per_epoch, num_epochs, power=0.9):
    """train_pse_cd.py:385-402 (Poly.get_lr, no warm-up): T = epoch*ipe + cur_iter."""
    T = epoch0 * iters_per_epoch + it_in_epoch
    return base_lr * (1 - T / (num_epochs * iters_per_epoch)) ** power
