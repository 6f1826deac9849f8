"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the pseudo-change pair synthesis (stcd_pseudo_pair).

PARITY UNPINNED: the reference assembles these pairs from files (/root/reference/data/dataset.py:468-482: B := in-painted
copy of A, label := building mask, or B := A, label := 0; ToTensor/Normalize :499-500; paired cutout :24-57) and holds
no generator arithmetic to check against.  What IS pinned to the reference here: the label rule (mask >= 1 -> 1,
dataset.py:461), the no-change branch (B == A, label 0), the normalisation constants and formula, and the cutout
contract (same values in A and B, label 255 inside the rectangle).  The blend inside the mask is this build's own spec.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
_M64 = (1 << 64) - 1


def _hash24(seed, i):
    z = (seed + 0x9E3779B97F4A7C15 * (i + 1)) & _M64          # splitmix64, as the kernel
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return ((z ^ (z >> 31)) >> 40) & 0xFFFFFF


def pseudo_pair(img_a, donor, mask, change, alpha=None, erase_xywh=None, seed=0, mean=MEAN, std=STD):
    """uint8 [B,H,W,3] x2, uint8 [B,H,W], flags [B] -> (x1, x2 fp32 [B,3,H,W], c_label, s_label_a, s_label_b int64 [B,H,W])."""
    B, H, W, _ = img_a.shape
    a = img_a.astype(np.float32)
    b = a.copy()
    m = mask >= 1
    al = np.ones(B, np.float32) if alpha is None else np.asarray(alpha, np.float32)
    for n in range(B):
        if change[n]:
            blend = np.rint(al[n] * donor[n].astype(np.float32) + (np.float32(1.0) - al[n]) * a[n])
            b[n][m[n]] = blend[m[n]]
    lab = m.astype(np.int64)
    c_label = np.where(np.asarray(change, bool)[:, None, None], lab, 0)
    s_a = lab.copy()
    s_b = np.where(np.asarray(change, bool)[:, None, None], 0, lab)
    if erase_xywh is not None:
        for n in range(B):
            ex, ey, ew, eh = (int(v) for v in erase_xywh[n])
            if ew <= 0 or eh <= 0:
                continue
            for y in range(max(ey, 0), min(ey + eh, H)):
                for x in range(max(ex, 0), min(ex + ew, W)):
                    i = (n * H + y) * W + x
                    for c in range(3):
                        v = float(_hash24(seed, i * 3 + c) & 255)
                        a[n, y, x, c] = v
                        b[n, y, x, c] = v
                    c_label[n, y, x] = 255
    inv = np.float32(1.0) / std.astype(np.float32)
    norm = lambda t: np.ascontiguousarray(((t * np.float32(1.0 / 255.0) - mean) * inv).transpose(0, 3, 1, 2)).astype(np.float32)
    return norm(a), norm(b), c_label, s_a, s_b
