"""On-device photometric augmentation of image pairs.

The reference augments on the host, per sample, through PIL / torchvision inside the DataLoader workers
(/root/reference/data/dataset.py:488-495): with probability 0.5 ``T.ColorJitter(0.5, 0.5, 0.5, 0.25)`` on A and on B
(one coin for the pair, independent factors), then per image ``T.RandomGrayscale(p=0.2)`` and ``blur(p=0.5)``
(GaussianBlur, sigma ~ U(0.1, 2.0), :120-124), then ToTensor + Normalize.  At the engine's throughput eight PIL workers
cannot feed one GPU, so the same recipe runs as HIP kernels on normalised batches that already sit in HBM
(``stcd_augment`` in include/stcd_hip.h).  ``draw_params`` reproduces the reference's probabilities and factor ranges;
the arithmetic per op is torchvision's float-tensor formula, in a FIXED order (ColorJitter permutes its four ops at
random: parity with a given torchvision run is not defined -- the per-op formulas are what the tests pin, to PIL).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import StcdError
from .pseudo import MEAN, STD


def draw_params(pairs: int, seed: int, jitter_p: float = 0.5, gray_p: float = 0.2, blur_p: float = 0.5) -> np.ndarray:
    """fp32 [2*pairs, 8] rows {jitter_on, brightness, contrast, saturation, hue, gray_on, sigma, 0}: rows [0, pairs) are the
    A images, rows [pairs, 2*pairs) the B images (dataset.py:488-495: one jitter coin per PAIR, independent factors)."""
    rng = np.random.default_rng(seed)
    p = np.zeros((2 * pairs, 8), np.float32)
    coin = rng.random(pairs) < jitter_p
    for half in range(2):
        sl = slice(half * pairs, (half + 1) * pairs)
        p[sl, 0] = coin
        p[sl, 1] = rng.uniform(0.5, 1.5, pairs)           # ColorJitter(brightness=0.5) -> U(max(0, 1-0.5), 1+0.5)
        p[sl, 2] = rng.uniform(0.5, 1.5, pairs)
        p[sl, 3] = rng.uniform(0.5, 1.5, pairs)
        p[sl, 4] = rng.uniform(-0.25, 0.25, pairs)        # hue=0.25
        p[sl, 5] = rng.random(pairs) < gray_p
        p[sl, 6] = np.where(rng.random(pairs) < blur_p, rng.uniform(0.1, 2.0, pairs), 0.0)
    return p


_SCRATCH = {}      # (device, bytes) -> scratch tensor: one allocation per shape, not per call
_PINNED = {}       # N -> pinned host staging buffer for the parameter upload


def _scratch_for(device, nbytes):
    key = (str(device), int(nbytes))
    t = _SCRATCH.get(key)
    if t is None:
        t = _SCRATCH[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return t


def _upload_params(params, n, device):
    """[N,8] fp32 on the device; host arrays go through a pinned staging buffer with a non-blocking copy (a pageable
    torch.as_tensor(...).to(device) synchronises the host with the stream every step)."""
    if torch.is_tensor(params):
        return params.to(device=device, dtype=torch.float32).contiguous()
    a = np.ascontiguousarray(np.asarray(params, np.float32))
    if a.shape != (n, 8):
        raise StcdError(f"params must be [N,8], got {a.shape}")
    pin = _PINNED.get(n)
    if pin is None:
        pin = _PINNED[n] = torch.empty((n, 8), dtype=torch.float32).pin_memory()
    ev = _PINNED.get(("ev", n))
    if ev is not None:
        ev.synchronize()                    # the previous upload from this staging buffer has left the host
    pin.copy_(torch.from_numpy(a))
    dev = pin.to(device, non_blocking=True)
    ev = _PINNED[("ev", n)] = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return dev


def augment(x: torch.Tensor, params, mean: Sequence[float] = MEAN, std: Sequence[float] = STD, out: Optional[torch.Tensor] = None):
    """x: normalised fp32 [N,3,H,W] on the GPU; params: [N,8] (numpy or tensor).  Returns the augmented, normalised batch."""
    if not x.is_cuda:
        raise StcdError("augment runs on the GPU (no CPU fallback)")
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32:
        raise StcdError(f"expected fp32 [N,3,H,W], got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    N, _, H, W = x.shape
    prm = _upload_params(params, N, x.device)
    if tuple(prm.shape) != (N, 8):
        raise StcdError(f"params must be [N,8], got {tuple(prm.shape)}")
    if out is None:
        out = torch.empty_like(x)
    elif (not out.is_cuda or out.device != x.device or out.dtype != torch.float32 or tuple(out.shape) != tuple(x.shape)
          or not out.is_contiguous() or out.data_ptr() == x.data_ptr()):
        raise StcdError(f"out must be a contiguous fp32 {tuple(x.shape)} tensor on {x.device}, distinct from x")
    l = _lib.lib()
    nb = l.stcd_augment_scratch_bytes(N, H, W)
    scratch = _scratch_for(x.device, nb)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    with torch.cuda.device(x.device):
        _lib.check(l.stcd_augment(C.c_void_p(x.data_ptr()), C.c_void_p(prm.data_ptr()), N, H, W, m3, s3, C.c_void_p(out.data_ptr()),
                                  C.c_void_p(scratch.data_ptr()), nb, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out


def augment_pair(x1: torch.Tensor, x2: torch.Tensor, seed: int):
    """The reference's train-time recipe on a batch of pairs: returns (x1', x2')."""
    b = x1.shape[0]
    y = augment(torch.cat([x1, x2], 0), draw_params(b, seed))
    return y[:b], y[b:]
