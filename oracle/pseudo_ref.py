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


# ------------------------------------------------------------------------------------------ photometric augmentation
def _gray(rgb):
    return np.float32(0.299) * rgb[0] + np.float32(0.587) * rgb[1] + np.float32(0.114) * rgb[2]


def _hue(rgb, hf):
    """torchvision.transforms._functional_tensor._rgb2hsv / _hsv2rgb on float images, hue shifted by hf (mod 1)."""
    r, g, b = rgb
    mx, mn = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    eqc = mx == mn
    cr = mx - mn
    sat = cr / np.where(eqc, np.float32(1), mx)
    crd = np.where(eqc, np.float32(1), cr)
    rc, gc, bc = (mx - r) / crd, (mx - g) / crd, (mx - b) / crd
    h = np.where(mx == r, bc - gc, 0) + np.where((mx == g) & (mx != r), 2 + rc - bc, 0) + np.where((mx != g) & (mx != r), 4 + gc - rc, 0)
    h = (h / 6 + 1) % 1
    h = (h + np.float32(hf)) % 1
    v = mx
    i6 = np.floor(h * 6)
    f = h * 6 - i6
    i = i6.astype(np.int64) % 6
    p = np.clip(v * (1 - sat), 0, 1); q = np.clip(v * (1 - sat * f), 0, 1); t = np.clip(v * (1 - sat * (1 - f)), 0, 1)
    out = np.empty_like(rgb)
    out[0] = np.choose(i, [v, q, p, p, t, v]); out[1] = np.choose(i, [t, v, v, q, p, p]); out[2] = np.choose(i, [p, p, t, v, v, q])
    return out.astype(np.float32)


def _blur1d(img, sigma, axis):
    rad = min(8, int(np.ceil(3.0 * sigma)))
    d = np.arange(-rad, rad + 1)
    w = np.exp(np.float32(-0.5 / (sigma * sigma)) * (d * d).astype(np.float32)).astype(np.float32)
    n = img.shape[axis]
    acc = np.zeros_like(img)
    for k, dd in enumerate(d):
        idx = np.clip(np.arange(n) + dd, 0, n - 1)
        acc += w[k] * np.take(img, idx, axis=axis)
    return acc / w.sum()


def augment(x, params, mean=MEAN, std=STD):
    """Restatement of stcd_augment: x normalised fp32 [N,3,H,W]; params [N,8] = {jitter_on, brightness, contrast,
    saturation, hue, gray_on, sigma, 0}.  Per-op formulas = torchvision F.adjust_* on float tensors
    (the reference calls them through T.ColorJitter / T.RandomGrayscale, data/dataset.py:488-495)."""
    x = np.asarray(x, np.float32)
    out = np.empty_like(x)
    m, s = mean.reshape(3, 1, 1), std.reshape(3, 1, 1)
    for n in range(x.shape[0]):
        on, br, ct, st, hf, gr, sigma, _ = (np.float32(v) for v in params[n])
        img = x[n] * s + m
        if on:
            img = np.clip(br * img, 0, 1)
            img = np.clip(ct * img + (1 - ct) * np.float32(_gray(img).mean(dtype=np.float64)), 0, 1)
            img = np.clip(st * img + (1 - st) * _gray(img)[None], 0, 1)
            if hf != 0:
                img = _hue(img, hf)
        if gr:
            img = np.repeat(_gray(img)[None], 3, 0)
        if sigma > 0:
            img = _blur1d(_blur1d(img.astype(np.float32), float(sigma), 2), float(sigma), 1)
        out[n] = (img - m) / s
    return out.astype(np.float32)
