"""stcd_pseudo_pair (on-device pseudo-change pair synthesis) against oracle/pseudo_ref.py, plus the properties the
reference's file-based assembly fixes (/root/reference/data/dataset.py:468-482, 499-500, 24-57)."""
import numpy as np
import pytest
import torch

from oracle import pseudo_ref as P
from stcd_amd import synth
from stcd_amd.pseudo import pseudo_change_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B, H, W, seed):
    rng = np.random.default_rng(seed)
    a, donor, lab = synth.make_pairs_u8(B, H, W, seed)
    mask = (lab * rng.integers(1, 256, size=lab.shape)).astype(np.uint8)      # any value >= 1 is "building"
    change = rng.integers(0, 2, size=B).astype(np.uint8)
    change[0], change[-1] = 1, 0
    return a, donor, mask, change, rng


@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (3, 33, 47), (4, 256, 256), (2, 512, 512)])
def test_pair_matches_oracle(B, H, W):
    a, donor, mask, change, rng = _case(B, H, W, 7 + H)
    alpha = rng.uniform(0.3, 1.0, size=B).astype(np.float32)
    alpha[0] = 1.0
    small = H * W <= 4096
    erase = np.zeros((B, 4), np.int32)
    if small:                                         # the oracle's erase loop is pure Python: small cases only
        erase[0] = (3, 2, 7, 5)
        erase[-1] = (W - 4, H - 3, 9, 9)              # clipped by the tile border
    t = lambda v: torch.from_numpy(v).to(DEV)
    got = pseudo_change_pairs(t(a), t(donor), t(mask), t(change), t(alpha), t(erase), seed=1234)
    want = P.pseudo_pair(a, donor, mask, change, alpha, erase, seed=1234)
    for g, w_, name in zip(got, want, ("x1", "x2", "c_label", "s_label_a", "s_label_b")):
        g = g.cpu().numpy()
        if g.dtype == np.float32:
            np.testing.assert_allclose(g, w_, rtol=0, atol=2e-6, err_msg=name)
        else:
            np.testing.assert_array_equal(g, w_, err_msg=name)


def test_reference_properties():
    """What dataset.py:468-482 / 499-500 / 24-57 fixes, whatever the blend: no-change tiles give B == A and label 0;
    alpha = 1 replaces exactly the masked pixels by the donor's; labels follow mask >= 1; x1 == synth normalisation."""
    B, H, W = 4, 64, 64
    a, donor, mask, change, _ = _case(B, H, W, 3)
    t = lambda v: torch.from_numpy(v).to(DEV)
    x1, x2, c, sa, sb = pseudo_change_pairs(t(a), t(donor), t(mask), t(change))
    x1, x2, c, sa, sb = (v.cpu().numpy() for v in (x1, x2, c, sa, sb))
    np.testing.assert_allclose(x1, synth.normalize_nchw(a), rtol=0, atol=2e-6)
    m = mask >= 1
    for n in range(B):
        if change[n]:
            want = np.where(m[n][..., None], donor[n], a[n])[None]
            np.testing.assert_allclose(x2[n:n + 1], synth.normalize_nchw(want), rtol=0, atol=2e-6)
            np.testing.assert_array_equal(c[n], m[n].astype(np.int64))
            assert not sb[n].any()
        else:
            np.testing.assert_array_equal(x2[n], x1[n])
            assert not c[n].any()
            np.testing.assert_array_equal(sb[n], m[n].astype(np.int64))
        np.testing.assert_array_equal(sa[n], m[n].astype(np.int64))


def test_generated_pairs_feed_the_engine():
    """config 4 of BASELINE.json in miniature: pseudo-change pairs at 512x512 straight into a SiamUnet_diff step."""
    from stcd_amd.losses import cross_entropy
    from stcd_amd.modules import SiamUnet_diff
    from stcd_amd.optim import FlatAdam
    a, donor, mask, change, _ = _case(2, 512, 512, 11)
    change[:] = 1
    t = lambda v: torch.from_numpy(v).to(DEV)
    x1, x2, c, _, _ = pseudo_change_pairs(t(a), t(donor), t(mask), t(change))
    m = SiamUnet_diff(3, 2).to(DEV).train()
    opt = FlatAdam(m, lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = cross_entropy(m(x1, x2), c)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_argument_checks():
    from stcd_amd._lib import StcdError
    z = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device=DEV)
    with pytest.raises(StcdError):
        pseudo_change_pairs(z.cpu(), z.cpu(), z[..., 0].cpu(), torch.ones(1))
    with pytest.raises(StcdError):
        pseudo_change_pairs(z.float(), z, z[..., 0], torch.ones(1))
    with pytest.raises(StcdError):
        pseudo_change_pairs(z, z, z[..., 0], torch.ones(2))
