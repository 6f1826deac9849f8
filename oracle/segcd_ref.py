"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, fp32) of the model the reference's scripts actually train:
``smp.SegCD(encoder_name="resnet50")`` (/root/reference/train_pse_cd.py:419-427, train_stcd.py:631-638), and of the same
class over the other plain ResNet encoders of the reference's registry (encoders/resnet.py:126-171: resnet18 / resnet34 with
BasicBlock, resnet101 / resnet152 with Bottleneck).

Restated from text (the ``segmentation_models_pytorch`` package cannot be imported here: its ``__init__`` pulls ``timm``;
``encoders/resnet.py`` pulls ``torchvision``):
  SegCD.__init__/forward     /root/reference/segmentation_models_pytorch/decoders/unet/model.py:267-332
  ResNetEncoder.get_stages   /root/reference/segmentation_models_pytorch/encoders/resnet.py:37-70  (resnet50: Bottleneck, [3,4,6,3],
                             out_channels (3, 64, 256, 512, 1024, 2048))
  ResNet / Bottleneck        /root/reference/models/resnet.py:78-124,127-190 (the torchvision code the encoder subclasses:
                             7x7 s2 stem, 3x3 s2 p1 max-pool, v1.5 bottlenecks with the stride on the 3x3, 1x1 s-strided
                             down-sample on the identity)
  BasicBlock                 /root/reference/models/resnet.py:37-75 (two 3x3 convs, stride on the first; a down-sample only
                             where the stride or the width changes, so layer1.0 has none)
  UnetDecoder / DecoderBlock /root/reference/segmentation_models_pytorch/decoders/unet/decoder.py:8-123
  Conv2dReLU                 /root/reference/segmentation_models_pytorch/base/modules.py:10-47 (bias-free conv + BN + ReLU)
  SegmentationHead           /root/reference/segmentation_models_pytorch/base/heads.py:5-11 (3x3 conv with bias)
PINNED by tests/golden/g10_segcd.npz, g11_segcd_2cls.npz (resnet50), g12_segcd_r18.npz, g13_segcd_r34.npz, g14_segcd_r101.npz
and g15_unetseg.npz (UnetSeg, model.py:109-171: the single-image twin train_sup.py:303 trains), g16_ffctlcd.npz (FFCTLCD,
model.py:335-423):
the reference's own ResNet (models/resnet.py), UnetDecoder (decoder.py, loaded as a file) and SegmentationHead assembled
exactly as SegCD.__init__ / forward state (tests/golden/make_golden.py:_segcd_fixture).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_MOMENTUM, BN_EPS = 0.1, 1e-5
LAYERS = (3, 4, 6, 3)                 # resnet50
PLANES = (64, 128, 256, 512)
ENC_OUT = (3, 64, 256, 512, 1024, 2048)
DEC_CH = (256, 128, 64, 32, 16)
# encoders/resnet.py:126-171: name -> (block expansion, blocks per stage)
ENCODERS = {"resnet18": (1, (2, 2, 2, 2)), "resnet34": (1, (3, 4, 6, 3)), "resnet50": (4, (3, 4, 6, 3)),
            "resnet101": (4, (3, 4, 23, 3)), "resnet152": (4, (3, 8, 36, 3))}


def enc_out(encoder="resnet50"):
    x = ENCODERS[encoder][0]
    return (3, 64, 64 * x, 128 * x, 256 * x, 512 * x)


def block_specs(encoder="resnet50"):
    """[(prefix, inplanes, width, stride, has_downsample)] in registration order (ResNet._make_layer, models/resnet.py:165-187:
    a down-sample where stride != 1 or inplanes != planes * expansion)."""
    x, layers = ENCODERS[encoder]
    out, inpl = [], 64
    for li, (nb, pl) in enumerate(zip(layers, PLANES)):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 0) else 1
            down = b == 0 and (stride != 1 or inpl != pl * x)
            out.append((f"encoder.layer{li + 1}.{b}", inpl, pl, stride, down))
            inpl = pl * x
    return out


def decoder_specs(encoder="resnet50"):
    """[(prefix, in_ch, skip_ch, out_ch)] (decoder.py:84-96)."""
    enc = list(enc_out(encoder)[1:])[::-1]              # resnet50: 2048, 1024, 512, 256, 64
    ins = [enc[0]] + list(DEC_CH[:-1])
    skips = enc[1:] + [0]
    return [(f"decoder.blocks.{i}", ins[i], skips[i], DEC_CH[i]) for i in range(5)]


def _bn_entries(name, c):
    return [(name + ".weight", (c,), "bn_w"), (name + ".bias", (c,), "bn_b"), (name + ".running_mean", (c,), "rm"),
            (name + ".running_var", (c,), "rv"), (name + ".num_batches_tracked", (), "nbt")]


def param_specs(in_ch=3, classes=1, encoder="resnet50"):
    """(name, shape, kind) in the reference's state_dict order."""
    x = ENCODERS[encoder][0]
    s = [("encoder.conv1.weight", (64, in_ch, 7, 7), "conv")] + _bn_entries("encoder.bn1", 64)
    for pre, inpl, w, stride, down in block_specs(encoder):
        if x == 4:
            s += [(pre + ".conv1.weight", (w, inpl, 1, 1), "conv")] + _bn_entries(pre + ".bn1", w)
            s += [(pre + ".conv2.weight", (w, w, 3, 3), "conv")] + _bn_entries(pre + ".bn2", w)
            s += [(pre + ".conv3.weight", (4 * w, w, 1, 1), "conv")] + _bn_entries(pre + ".bn3", 4 * w)
        else:
            s += [(pre + ".conv1.weight", (w, inpl, 3, 3), "conv")] + _bn_entries(pre + ".bn1", w)
            s += [(pre + ".conv2.weight", (w, w, 3, 3), "conv")] + _bn_entries(pre + ".bn2", w)
        if down:
            s += [(pre + ".downsample.0.weight", (x * w, inpl, 1, 1), "conv")] + _bn_entries(pre + ".downsample.1", x * w)
    for pre, cin, cskip, cout in decoder_specs(encoder):
        s += [(pre + ".conv1.0.weight", (cout, cin + cskip, 3, 3), "conv")] + _bn_entries(pre + ".conv1.1", cout)
        s += [(pre + ".conv2.0.weight", (cout, cout, 3, 3), "conv")] + _bn_entries(pre + ".conv2.1", cout)
    s += [("segmentation_head.0.weight", (classes, DEC_CH[-1], 3, 3), "conv"), ("segmentation_head.0.bias", (classes,), "bias")]
    return s


def synth_state(in_ch=3, classes=1, seed=0, perturb_running=False, encoder="resnet50"):
    """Deterministic state dict with the reference's names / shapes (weights are NOT the reference's init: He-scaled normals so
    activations stay O(1) through 50 layers; BN gamma ~ 1, beta small)."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shape, kind in param_specs(in_ch, classes, encoder):
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            st[name] = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        elif kind == "bias":
            st[name] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif kind == "bn_w":
            st[name] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif kind == "bn_b":
            st[name] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif kind == "rm":
            st[name] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32)) if perturb_running else torch.zeros(shape)
        elif kind == "rv":
            st[name] = torch.from_numpy((1.0 + 0.2 * np.abs(rng.standard_normal(shape))).astype(np.float32)) if perturb_running else torch.ones(shape)
        else:
            st[name] = torch.zeros((), dtype=torch.int64)
    return st


def _bn(x, st, name, training):
    g, b, rm, rv = st[name + ".weight"], st[name + ".bias"], st[name + ".running_mean"], st[name + ".running_var"]
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
            st[name + ".num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]


def _encoder_of(st):
    """Which registry entry a state dict belongs to (block type from layer1.0, depths from the key list)."""
    basic = "encoder.layer1.0.conv3.weight" not in st
    depth = tuple(sum(1 for k in st if k.startswith(f"encoder.layer{li}.") and k.endswith(".conv1.weight")) for li in (1, 2, 3, 4))
    for name, (x, layers) in ENCODERS.items():
        if (x == 1) == basic and tuple(layers) == depth:
            return name
    raise KeyError(f"no ResNet encoder with blocks {depth}, basic={basic}")


def encoder(st, x, training):
    """ResNetEncoder.forward: features [x, relu(bn1(conv1 x)), layer1(maxpool .), layer2, layer3, layer4]."""
    name = _encoder_of(st)
    xp, layers = ENCODERS[name]
    feats = [x]
    x = torch.relu(_bn(F.conv2d(x, st["encoder.conv1.weight"], None, stride=2, padding=3), st, "encoder.bn1", training))
    feats.append(x)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    specs = block_specs(name)
    ends = np.cumsum(layers)
    for i, (pre, inpl, w, stride, down) in enumerate(specs):
        idt = x
        if xp == 4:
            out = torch.relu(_bn(F.conv2d(x, st[pre + ".conv1.weight"]), st, pre + ".bn1", training))
            out = torch.relu(_bn(F.conv2d(out, st[pre + ".conv2.weight"], None, stride=stride, padding=1), st, pre + ".bn2", training))
            out = _bn(F.conv2d(out, st[pre + ".conv3.weight"]), st, pre + ".bn3", training)
        else:       # BasicBlock (models/resnet.py:57-75)
            out = torch.relu(_bn(F.conv2d(x, st[pre + ".conv1.weight"], None, stride=stride, padding=1), st, pre + ".bn1", training))
            out = _bn(F.conv2d(out, st[pre + ".conv2.weight"], None, padding=1), st, pre + ".bn2", training)
        if down:
            idt = _bn(F.conv2d(x, st[pre + ".downsample.0.weight"], None, stride=stride), st, pre + ".downsample.1", training)
        x = torch.relu(out + idt)
        if i + 1 in ends:
            feats.append(x)
    return feats


def decoder(st, feats, training):
    """UnetDecoder.forward (decoder.py:108-123): nearest x2, cat skip, two Conv2dReLU; center = Identity for resnets."""
    feats = feats[1:][::-1]
    x, skips = feats[0], feats[1:]
    for i, (pre, cin, cskip, cout) in enumerate(decoder_specs(_encoder_of(st))):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i < len(skips):
            x = torch.cat([x, skips[i]], dim=1)
        x = torch.relu(_bn(F.conv2d(x, st[pre + ".conv1.0.weight"], None, padding=1), st, pre + ".conv1.1", training))
        x = torch.relu(_bn(F.conv2d(x, st[pre + ".conv2.0.weight"], None, padding=1), st, pre + ".conv2.1", training))
    return x


def forward(st, A, B, training=False):
    """SegCD.forward (model.py:316-332): (mask_t1, mask_t2, change = min(head(|d1 - d2|), |m1 - m2|)).
    The shared encoder / decoder BatchNorms see date A, then date B (two separate batch-stat normalisations)."""
    d1 = decoder(st, encoder(st, A, training), training)
    d2 = decoder(st, encoder(st, B, training), training)
    head = lambda t: F.conv2d(t, st["segmentation_head.0.weight"], st["segmentation_head.0.bias"], padding=1)
    m1, m2 = head(d1), head(d2)
    diffea = head(torch.abs(d1 - d2))
    diffseg = torch.abs(m1 - m2)
    return m1, m2, torch.min(diffea, diffseg)


def unetseg_forward(st, x, training=False):
    """UnetSeg.forward (model.py:165-171): masks = head(decoder(*encoder(x))) -- one image batch, one BatchNorm call per layer."""
    d = decoder(st, encoder(st, x, training), training)
    return F.conv2d(d, st["segmentation_head.0.weight"], st["segmentation_head.0.bias"], padding=1)


def ffctlcd_forward(st, A, B, training=False):
    """FFCTLCD.forward (model.py:407-423): the decoder + head on |f1 - f2| FIRST, then on each date's features (three BatchNorm
    calls per decoder layer, in that order); change = min(head(dec(|f1 - f2|)), |mask_t1 - mask_t2|)."""
    f1, f2 = encoder(st, A, training), encoder(st, B, training)
    head = lambda t: F.conv2d(t, st["segmentation_head.0.weight"], st["segmentation_head.0.bias"], padding=1)
    diffea = head(decoder(st, [torch.abs(a - b) for a, b in zip(f1, f2)], training))
    m1 = head(decoder(st, f1, training))
    m2 = head(decoder(st, f2, training))
    return m1, m2, torch.min(diffea, torch.abs(m1 - m2))
