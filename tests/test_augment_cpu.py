"""The augmentation oracle (oracle/pseudo_ref.py:augment) against PIL -- the library torchvision's T.ColorJitter /
T.RandomGrayscale / the reference's blur() call into for PIL images (/root/reference/data/dataset.py:120-124,488-495).
PIL works on uint8, the oracle on floats: agreement to the uint8 quantisation (stated per op)."""
import numpy as np
from PIL import Image, ImageEnhance, ImageFilter

from oracle import pseudo_ref as P

ID = dict(mean=np.zeros(3, np.float32), std=np.ones(3, np.float32))


def _img(seed=0, h=40, w=48):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h // 4, w // 4, 3)).astype(np.uint8)
    return np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR))


def _run(u8, **kw):
    p = np.zeros((1, 8), np.float32)
    p[0, 1:4] = 1.0
    for k, v in kw.items():
        p[0, {"on": 0, "b": 1, "c": 2, "s": 3, "h": 4, "gray": 5, "sigma": 6}[k]] = v
    x = (u8.astype(np.float32) / 255.0).transpose(2, 0, 1)[None]
    return P.augment(x, p, **ID)[0].transpose(1, 2, 0) * 255.0


def test_identity_and_brightness_contrast_saturation_match_pil_enhance():
    u8 = _img()
    np.testing.assert_allclose(_run(u8), u8, atol=1e-3)
    pil = Image.fromarray(u8)
    for f in (0.5, 0.8, 1.3, 1.5):      # F_pil.adjust_brightness / _contrast / _saturation == ImageEnhance.*(img).enhance(f)
        np.testing.assert_allclose(_run(u8, on=1, b=f), np.asarray(ImageEnhance.Brightness(pil).enhance(f), np.float32), atol=1.01)
        np.testing.assert_allclose(_run(u8, on=1, c=f), np.asarray(ImageEnhance.Contrast(pil).enhance(f), np.float32), atol=1.6)
        np.testing.assert_allclose(_run(u8, on=1, s=f), np.asarray(ImageEnhance.Color(pil).enhance(f), np.float32), atol=1.6)


def test_grayscale_matches_pil_luma():
    u8 = _img(1)
    got = _run(u8, gray=1)
    ref = np.asarray(Image.fromarray(u8).convert("L"), np.float32)
    for c in range(3):
        np.testing.assert_allclose(got[..., c], ref, atol=1.0)


def test_hue_matches_pil_hsv_shift():
    """F_pil.adjust_hue: convert('HSV'), h += uint8(hf * 255) (wrapping), back to RGB -- 8-bit HSV: a few grey levels."""
    u8 = _img(2)
    for hf in (-0.25, -0.1, 0.07, 0.25):
        h, s, v = Image.fromarray(u8).convert("HSV").split()
        hh = (np.asarray(h, np.uint8).astype(np.int32) + int(np.uint8(np.int32(hf * 255)))) % 256      # uint8 arithmetic wraps
        ref = np.asarray(Image.merge("HSV", (Image.fromarray(hh.astype(np.uint8), "L"), s, v)).convert("RGB"), np.float32)
        got = _run(u8, on=1, h=hf)
        assert np.abs(got - ref).mean() < 3.0, (hf, np.abs(got - ref).mean())
        assert np.percentile(np.abs(got - ref), 99) < 12.0


def test_gaussian_blur_tracks_pil():
    """PIL approximates GaussianBlur(radius=sigma) by repeated box filters; the oracle is the exact separable Gaussian."""
    u8 = _img(3, 64, 64)
    for sigma in (0.6, 1.2, 2.0):
        ref = np.asarray(Image.fromarray(u8).filter(ImageFilter.GaussianBlur(radius=sigma)), np.float32)
        got = _run(u8, sigma=sigma)
        assert np.abs(got - ref)[4:-4, 4:-4].mean() < 2.5, (sigma, np.abs(got - ref)[4:-4, 4:-4].mean())
    flat = np.full((16, 16, 3), 77, np.uint8)
    np.testing.assert_allclose(_run(flat, sigma=1.5), 77.0, atol=1e-3)      # a constant image is a fixed point


def test_draw_params_follow_the_reference_probabilities():
    from stcd_amd.augment import draw_params
    p = draw_params(4000, seed=1)
    a, b = p[:4000], p[4000:]
    assert np.array_equal(a[:, 0], b[:, 0])                         # one jitter coin per pair (dataset.py:488-490)
    assert abs(a[:, 0].mean() - 0.5) < 0.03 and abs(p[:, 5].mean() - 0.2) < 0.02 and abs((p[:, 6] > 0).mean() - 0.5) < 0.03
    assert p[:, 1:4].min() >= 0.5 and p[:, 1:4].max() <= 1.5 and np.abs(p[:, 4]).max() <= 0.25
    s = p[p[:, 6] > 0, 6]
    assert s.min() >= 0.1 and s.max() <= 2.0
    assert not np.array_equal(a[:, 1], b[:, 1])                     # independent factors for A and B
