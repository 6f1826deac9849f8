"""On-device pseudo-change pair synthesis.

The reference builds a pseudo-change sample from files (/root/reference/data/dataset.py:468-482): image B is an
in-painted copy of A (``WHU-A/*.png``) and the change label is A's building mask when the tile is in the change list;
otherwise B is A and the label is all zero; both images then go through ToTensor + Normalize (:499-500).  It contains no
generator arithmetic, so ``stcd_pseudo_pair`` (include/stcd_hip.h) defines one: inside the mask B is
``round(alpha * donor + (1 - alpha) * A)`` (alpha = 1 reproduces "replace by the donor / in-painted pixels"), outside
it B is A; the optional paired cutout (dataset.py:24-57) erases the same rectangle in A and B with per-pixel uniform
values and marks it 255 in the label.  uint8 HWC tiles in, normalised fp32 NCHW pair + int64 labels out, ready for
``model(x1, x2)``: the PIL/ToTensor/Normalize host path is gone.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import StcdError

MEAN = (0.485, 0.456, 0.406)     # dataset.py:129-130
STD = (0.229, 0.224, 0.225)


def pseudo_change_pairs(img_a: torch.Tensor, donor: torch.Tensor, mask: torch.Tensor, change: torch.Tensor,
                        alpha: Optional[torch.Tensor] = None, erase_xywh: Optional[torch.Tensor] = None, seed: int = 0,
                        mean: Sequence[float] = MEAN, std: Sequence[float] = STD):
    """img_a, donor: uint8 [B,H,W,3]; mask: uint8 [B,H,W]; change: bool/uint8 [B]; alpha: fp32 [B] or None;
    erase_xywh: int32 [B,4] or None.  All on one GPU.  Returns (x1, x2, c_label, s_label_a, s_label_b)."""
    if not img_a.is_cuda:
        raise StcdError("pseudo_change_pairs runs on the GPU (no CPU fallback)")
    if img_a.dtype != torch.uint8 or donor.dtype != torch.uint8 or mask.dtype != torch.uint8:
        raise StcdError("img_a, donor and mask must be uint8")
    if img_a.dim() != 4 or img_a.shape[-1] != 3 or donor.shape != img_a.shape or tuple(mask.shape) != tuple(img_a.shape[:3]):
        raise StcdError(f"expected uint8 [B,H,W,3] images and a [B,H,W] mask, got {tuple(img_a.shape)}, {tuple(donor.shape)}, {tuple(mask.shape)}")
    B, H, W, _ = img_a.shape
    dev = img_a.device
    img_a, donor, mask = img_a.contiguous(), donor.contiguous(), mask.contiguous()
    change = change.to(device=dev, dtype=torch.uint8).contiguous()
    if change.numel() != B:
        raise StcdError("change must hold one flag per sample")
    if alpha is not None:
        alpha = alpha.to(device=dev, dtype=torch.float32).contiguous()
    if erase_xywh is not None:
        erase_xywh = erase_xywh.to(device=dev, dtype=torch.int32).contiguous()
        if tuple(erase_xywh.shape) != (B, 4):
            raise StcdError("erase_xywh must be int32 [B,4]")
    x1 = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    x2 = torch.empty_like(x1)
    c_label = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    s_a, s_b = torch.empty_like(c_label), torch.empty_like(c_label)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().stcd_pseudo_pair(ptr(img_a), ptr(donor), ptr(mask), ptr(change), ptr(alpha), ptr(erase_xywh),
                                               C.c_uint64(seed & (2 ** 64 - 1)), B, H, W, m3, s3, ptr(x1), ptr(x2), ptr(c_label),
                                               ptr(s_a), ptr(s_b), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return x1, x2, c_label, s_a, s_b
