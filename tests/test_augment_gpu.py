"""stcd_augment (HIP) against its numpy restatement (oracle/pseudo_ref.py:augment, itself pinned to PIL on the CPU) and
through properties at the bench size."""
import numpy as np
import pytest
import torch

from oracle import pseudo_ref as P
from stcd_amd import augment as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _norm_batch(n, h, w, seed):
    rng = np.random.default_rng(seed)
    u = rng.random((n, 3, h // 2, w // 2)).astype(np.float32)
    u = np.repeat(np.repeat(u, 2, 2), 2, 3)[:, :, :h, :w]
    return ((u - P.MEAN.reshape(1, 3, 1, 1)) / P.STD.reshape(1, 3, 1, 1)).astype(np.float32)


@pytest.mark.parametrize("n,h,w", [(6, 32, 40), (4, 17, 23), (2, 64, 64)])
def test_augment_matches_oracle(n, h, w):
    x = _norm_batch(n, h, w, n + h)
    prm = A.draw_params(n // 2, seed=3 + n)
    prm[0, 0] = 1; prm[0, 4] = 0.2; prm[0, 6] = 1.3                  # make sure every branch is taken at least once
    prm[1, 0] = 0; prm[1, 5] = 1; prm[1, 6] = 0
    got = A.augment(torch.from_numpy(x).to(DEV), prm).cpu().numpy()
    ref = P.augment(x, prm)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=3e-4)


def test_augment_properties_at_bench_size():
    """32 images of 256x256 (16 pairs): identity parameters return the input; grayscale makes the three de-normalised
    channels equal; blurring a constant image changes nothing; pairs share the jitter coin; output finite."""
    n, h, w = 32, 256, 256
    x = torch.from_numpy(_norm_batch(n, h, w, 5)).to(DEV)
    ident = np.zeros((n, 8), np.float32); ident[:, 1:4] = 1
    np.testing.assert_allclose(A.augment(x, ident).cpu().numpy(), x.cpu().numpy(), rtol=0, atol=2e-6)
    g = ident.copy(); g[:, 5] = 1
    y = A.augment(x, g).cpu().numpy() * P.STD.reshape(1, 3, 1, 1) + P.MEAN.reshape(1, 3, 1, 1)
    assert np.abs(y[:, 0] - y[:, 1]).max() < 1e-5 and np.abs(y[:, 0] - y[:, 2]).max() < 1e-5
    const = torch.zeros_like(x) + 0.3
    b = ident.copy(); b[:, 6] = 1.7
    np.testing.assert_allclose(A.augment(const, b).cpu().numpy(), const.cpu().numpy(), atol=1e-5)
    y1, y2 = A.augment_pair(x[:16], x[16:], seed=9)
    assert torch.isfinite(y1).all() and torch.isfinite(y2).all() and y1.shape == x[:16].shape
    z1, z2 = A.augment_pair(x[:16], x[16:], seed=9)
    assert torch.equal(y1, z1) and torch.equal(y2, z2)              # same seed, same batch: reproducible
