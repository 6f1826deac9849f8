"""Synthetic LEVIR-CD-shaped bi-temporal pairs (no dataset ships with the build).

Shape/dtype contract follows the reference loaders: RGB tiles scaled to [0,1] then
normalised with the ImageNet mean/std (/root/reference/data/dataset.py:426-427,499-500),
labels in {0,1} (dataset.py:542-545).  The *content* is ours:
smoothed uniform noise for image A; image B is A with 0..3 axis-aligned rectangles
(1-8 % of the tile each) re-textured; the label is the union of those rectangles.
numpy only -- usable on the host with or without a GPU.
"""
from __future__ import annotations

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _box3(img: np.ndarray) -> np.ndarray:
    """3x3 box filter with edge replication, img [N,H,W,C] float32."""
    p = np.pad(img, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="edge")
    h, w = img.shape[1], img.shape[2]
    acc = np.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            acc += p[:, dy:dy + h, dx:dx + w, :]
    return acc / 9.0


def make_pairs_u8(n: int, h: int = 256, w: int = 256, seed: int = 1337):
    """-> (A u8 [n,h,w,3], B u8 [n,h,w,3], label u8 [n,h,w] in {0,1})."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, h, w, 3)).astype(np.float32)
    a = _box3(_box3(a))
    # stretch contrast back out after smoothing so the tiles are not flat grey
    a = np.clip((a - 127.5) * 4.0 + 127.5, 0, 255)
    b = a.copy()
    label = np.zeros((n, h, w), dtype=np.uint8)
    for i in range(n):
        for _ in range(int(rng.integers(0, 4))):
            area = rng.uniform(0.01, 0.08) * h * w
            aspect = rng.uniform(0.5, 2.0)
            rh = int(np.clip(round(np.sqrt(area * aspect)), 2, h))
            rw = int(np.clip(round(area / max(rh, 1)), 2, w))
            y0 = int(rng.integers(0, h - rh + 1))
            x0 = int(rng.integers(0, w - rw + 1))
            tex = rng.integers(0, 256, size=(1, rh, rw, 3)).astype(np.float32)
            tex = np.clip((_box3(tex) - 127.5) * 2.0 + rng.uniform(60, 200), 0, 255)
            b[i, y0:y0 + rh, x0:x0 + rw, :] = tex[0]
            label[i, y0:y0 + rh, x0:x0 + rw] = 1
    return a.astype(np.uint8), b.astype(np.uint8), label


def normalize_nchw(img_u8: np.ndarray) -> np.ndarray:
    """uint8 [n,h,w,3] -> fp32 [n,3,h,w], ((x/255) - mean) / std."""
    x = img_u8.astype(np.float32) / 255.0
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def make_batch(n: int, h: int = 256, w: int = 256, seed: int = 1337):
    """-> (A fp32 [n,3,h,w], B fp32 [n,3,h,w], label int64 [n,h,w])."""
    a, b, lab = make_pairs_u8(n, h, w, seed)
    return normalize_nchw(a), normalize_nchw(b), lab.astype(np.int64)
