"""``SegCD``: the ResNet-50 UNet change detector the reference's scripts train
(``smp.SegCD(encoder_name="resnet50", encoder_weights="imagenet")``, /root/reference/train_pse_cd.py:419-427,
train_stcd.py:631-638), as a drop-in module over the HIP engine -- and the same class over the other plain ResNet encoders of
the reference's registry (encoders/resnet.py:126-171): resnet18 / resnet34 (BasicBlock, models/resnet.py:37-75), resnet101 /
resnet152 (Bottleneck, :78-124).

Same constructor arguments, ``forward(A, B) -> (mask_t1, mask_t2, change)`` contract and ``state_dict`` keys / shapes as
/root/reference/segmentation_models_pytorch/decoders/unet/model.py:267-332 (encoder: encoders/resnet.py:37-70 over
models/resnet.py:78-190 with ``fc`` / ``avgpool`` removed; decoder: decoders/unet/decoder.py:8-123; head: base/heads.py:5-11),
so checkpoints interchange both ways.  The sub-modules are parameter holders only -- they are never called; forward and
backward are single calls into libstcd_hip.so (both dates batched through the shared encoder / decoder, per-date
BatchNorm statistics).  No CPU fallback.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from ._lib import StcdError
from .modules import HipChangeDetector

_PLANES = (64, 128, 256, 512)
# encoders/resnet.py:126-171 ("params" of each registry entry): name -> (block expansion, blocks per stage), and the file
# `encoder_weights="imagenet"` names for it (pretrainedmodels' torchvision URLs)
_ENCODERS = {"resnet18": (1, (2, 2, 2, 2), "resnet18-5c106cde.pth"), "resnet34": (1, (3, 4, 6, 3), "resnet34-333f7ec4.pth"),
             "resnet50": (4, (3, 4, 6, 3), "resnet50-19c8e357.pth"), "resnet101": (4, (3, 4, 23, 3), "resnet101-5d3b4d8f.pth"),
             "resnet152": (4, (3, 8, 36, 3), "resnet152-b121ed2d.pth")}
_URL_ROOT = "https://download.pytorch.org/models/"


def _enc_out(name):
    x = _ENCODERS[name][0]
    return (3, 64, 64 * x, 128 * x, 256 * x, 512 * x)


class _Bottleneck(nn.Module):
    """torchvision-style v1.5 bottleneck (models/resnet.py:78-124): stride on the 3x3."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class _BasicBlock(nn.Module):
    """BasicBlock holders (models/resnet.py:37-75): two 3x3 convs, the stride on the first."""
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class _ResNetEncoder(nn.Module):
    """ResNetEncoder holders (encoders/resnet.py:37-70) for one registry entry."""

    def __init__(self, name, in_channels):
        super().__init__()
        x, layers, _ = _ENCODERS[name]
        self.out_channels = (in_channels,) + _enc_out(name)[1:]
        self._depth, self._in_channels = 5, in_channels
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inpl = 64
        for li, (nb, pl) in enumerate(zip(layers, _PLANES)):
            blocks = []
            for b in range(nb):
                stride = 2 if (b == 0 and li > 0) else 1
                down = None
                if b == 0 and (stride != 1 or inpl != pl * x):           # ResNet._make_layer (models/resnet.py:165-187)
                    down = nn.Sequential(nn.Conv2d(inpl, pl * x, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(pl * x))
                blocks.append((_Bottleneck if x == 4 else _BasicBlock)(inpl, pl, stride, down))
                inpl = pl * x
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        for m in self.modules():                      # models/resnet.py:157-163
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class _DecoderBlock(nn.Module):
    """DecoderBlock holders (decoders/unet/decoder.py:8-46): two bias-free Conv2dReLU (conv, BN, ReLU)."""

    def __init__(self, in_ch, skip_ch, out_ch):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_ch + skip_ch, out_ch, 3, padding=1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))
        self.attention1 = nn.Identity()
        self.conv2 = nn.Sequential(nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))
        self.attention2 = nn.Identity()


class _UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        ins = [enc[0]] + list(decoder_channels[:-1])
        skips = enc[1:] + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList([_DecoderBlock(i, s, o) for i, s, o in zip(ins, skips, decoder_channels)])
        for m in self.modules():                      # base/initialization.py:4-20
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class SegCD(HipChangeDetector):
    """``SegCD(encoder_name="resnet34", encoder_weights=..., in_channels=3, classes=1).forward(A, B)`` ->
    ``(mask_t1, mask_t2, change)`` with ``change = min(head(|d1 - d2|), |mask_t1 - mask_t2|)`` (model.py:316-332).

    Supported configurations: the one the scripts use (resnet50) and the other plain ResNet encoders of the registry
    (resnet18 / 34 / 101 / 152); depth 5, BatchNorm decoder (256,128,64,32,16), no attention, no activation, no aux head;
    anything else raises NotImplementedError.  ``encoder_weights``: None (the encoder's own random init), a path to that
    ResNet's ``state_dict`` file, or "imagenet" (fetched through torch.hub like the reference does: needs the file in the hub
    cache when there is no network).  H and W must be multiples of 32."""

    ARCH = "segcd"
    FAMILY = "segcd"
    OUT_MAPS = 3

    def __init__(self, encoder_name: str = "resnet34", encoder_depth: int = 5, encoder_weights: Optional[str] = None,
                 decoder_use_batchnorm: bool = True, decoder_channels: List[int] = (256, 128, 64, 32, 16),
                 decoder_attention_type: Optional[str] = None, in_channels: int = 3, classes: int = 1, activation=None,
                 aux_params: Optional[dict] = None, dtype: Optional[str] = None):
        # Defaults: encoder_name "resnet34" as the reference's SegCD (decoders/unet/model.py:270-272) and the two sibling classes, so a
        # bare or positional call ported from the reference builds the same network (the scripts pass "resnet50" explicitly,
        # train_pse_cd.py:426).  encoder_weights defaults to None, NOT the reference's "imagenet": a deliberate deviation -- there is
        # no network here to fetch the checkpoint from; pass "imagenet" (hub cache) or a path to load it.
        if (encoder_name not in _ENCODERS or encoder_depth != 5 or decoder_use_batchnorm is not True or tuple(decoder_channels) != (256, 128, 64, 32, 16)
                or decoder_attention_type is not None or activation is not None or aux_params is not None):
            raise NotImplementedError("SegCD on the HIP engine: resnet18 / 34 / 50 / 101 / 152 encoder, depth 5, BatchNorm decoder "
                                      "(256,128,64,32,16), no attention / activation / aux head (train_pse_cd.py:426 builds resnet50)")
        if not 1 <= in_channels <= 8:
            raise NotImplementedError("SegCD on the HIP engine: 1..8 input channels")
        self.ARCH = self.FAMILY + "_" + encoder_name
        super().__init__(in_channels, classes, dtype)
        self.encoder = _ResNetEncoder(encoder_name, in_channels)
        self.inchannels = in_channels
        self.encoder_channels = self.encoder.out_channels
        self.decoder = _UnetDecoder(self.encoder.out_channels, tuple(decoder_channels))
        self.segmentation_head = nn.Sequential(nn.Conv2d(decoder_channels[-1], classes, kernel_size=3, padding=1), nn.Identity(), nn.Identity())
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)      # base/initialization.py:22-27
        nn.init.constant_(self.segmentation_head[0].bias, 0)
        self.name = "u-{}".format(encoder_name)
        self._ctor = dict(encoder_name=encoder_name, in_channels=in_channels, classes=classes)
        if encoder_weights is not None:
            self._load_encoder(encoder_weights)
        self._check_layout()

    def _load_encoder(self, weights: str):
        if os.path.exists(weights):
            sd = torch.load(weights, map_location="cpu")
        elif weights == "imagenet":
            sd = torch.hub.load_state_dict_from_url(_URL_ROOT + _ENCODERS[self._ctor["encoder_name"]][2], map_location="cpu")
        else:
            raise KeyError("Wrong pretrained weights `{}` for encoder `{}`. Available options are: "
                           "['imagenet', <path to a state_dict file>]".format(weights, self._ctor["encoder_name"]))
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
        if self.inchannels != 3:          # encoders/_utils.py patch_first_conv: 1 channel = sum, otherwise cycle RGB scaled by 3/C
            w = sd["conv1.weight"]
            if self.inchannels == 1:
                w = w.sum(1, keepdim=True)
            else:
                w = torch.stack([w[:, i % 3] for i in range(self.inchannels)], 1) * (3.0 / self.inchannels)
            sd["conv1.weight"] = w
        self.encoder.load_state_dict(sd)

    def __deepcopy__(self, memo):
        new = type(self)(dtype=self._engine.dtype, **self._ctor)
        new.load_state_dict({k: v.detach().clone() for k, v in self.state_dict().items()})
        new.train(self.training)
        if self._flat_params is not None:
            new.to(self._flat_params.device)
        return new

    def forward(self, A, B):
        if A.dim() == 4 and (A.shape[2] % 32 or A.shape[3] % 32):
            raise StcdError(f"SegCD needs H and W divisible by 32, got {tuple(A.shape[2:])} "
                            "(the reference's torch.cat of the x2 up-sampled maps fails for other sizes)")
        return super().forward(A, B)

    def _wrap_output(self, out, B):
        if isinstance(out, tuple):          # training: the autograd node already returns the three maps
            return out
        return out[:B], out[B:2 * B], out[2 * B:]



class UnetSeg(SegCD):
    """``UnetSeg(encoder_name=..., in_channels=3, classes=1).forward(x)`` -> masks: the single-image ResNet UNet
    ``train_sup.py:303`` trains (``smp.UnetSeg(encoder_name="resnet50", encoder_weights="imagenet")``;
    /root/reference/segmentation_models_pytorch/decoders/unet/model.py:109-171) -- SegCD's encoder / decoder / head (same
    ``state_dict``, so a supervised checkpoint loads into SegCD and back) on ONE image batch, BatchNorm over the whole batch."""

    FAMILY = "unetseg"
    OUT_MAPS = 1

    def __init__(self, encoder_name: str = "resnet34", encoder_depth: int = 5, encoder_weights: Optional[str] = None,
                 decoder_use_batchnorm: bool = True, decoder_channels: List[int] = (256, 128, 64, 32, 16),
                 decoder_attention_type: Optional[str] = None, in_channels: int = 3, classes: int = 1, activation=None,
                 aux_params: Optional[dict] = None, dtype: Optional[str] = None):
        super().__init__(encoder_name, encoder_depth, encoder_weights, decoder_use_batchnorm, decoder_channels, decoder_attention_type,
                         in_channels, classes, activation, aux_params, dtype)

    def forward(self, x):
        return super().forward(x, x)           # the engine reads the first input only (STCD_ARCH_UNETSEG)

    def _wrap_output(self, out, B):
        return out


class FFCTLCD(SegCD):
    """``FFCTLCD(encoder_name=...).forward(A, B)`` -> ``(mask_t1, mask_t2, change)``: the feature-level variant kept as a commented
    alternative in the scripts (``smp.FFCTLCD(encoder_name="resnet50", ...)``, train_pse_cd.py:419, train_stcd.py:637;
    /root/reference/segmentation_models_pytorch/decoders/unet/model.py:335-423).  The shared decoder + head also run on the
    encoder features' ``|f1 - f2|`` -- three decoder passes per forward (difference, date 1, date 2: each with its own BatchNorm
    batch statistics, running statistics updated in that order) -- and ``change = min(head(dec(|f1 - f2|)), |mask_t1 - mask_t2|)``.
    Same ``state_dict`` as SegCD / UnetSeg."""

    FAMILY = "ffctlcd"
    OUT_MAPS = 3

    def __init__(self, encoder_name: str = "resnet34", encoder_depth: int = 5, encoder_weights: Optional[str] = None,
                 decoder_use_batchnorm: bool = True, decoder_channels: List[int] = (256, 128, 64, 32, 16),
                 decoder_attention_type: Optional[str] = None, in_channels: int = 3, classes: int = 1, activation=None,
                 aux_params: Optional[dict] = None, dtype: Optional[str] = None):
        super().__init__(encoder_name, encoder_depth, encoder_weights, decoder_use_batchnorm, decoder_channels, decoder_attention_type,
                         in_channels, classes, activation, aux_params, dtype)
