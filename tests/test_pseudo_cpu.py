"""CPU checks of the pseudo-change oracle itself (oracle/pseudo_ref.py): the properties the reference's file-based
assembly fixes (/root/reference/data/dataset.py:468-482, 499-500, 24-57)."""
import numpy as np

from oracle import pseudo_ref as P
from stcd_amd import synth


def test_oracle_properties():
    a, donor, lab = synth.make_pairs_u8(3, 24, 20, seed=5)
    mask = (lab * 200).astype(np.uint8)
    change = np.array([1, 0, 1], np.uint8)
    erase = np.array([[2, 3, 5, 4], [0, 0, 0, 0], [18, 22, 6, 6]], np.int32)
    x1, x2, c, sa, sb = P.pseudo_pair(a, donor, mask, change, None, erase, seed=9)
    assert x1.shape == (3, 3, 24, 20) and x1.dtype == np.float32 and c.dtype == np.int64
    np.testing.assert_array_equal(x2[1], x1[1])                 # no-change tile: B == A, label 0
    assert not c[1].any()
    np.testing.assert_array_equal(sb[1], (mask[1] >= 1).astype(np.int64))
    assert (c[0, 3:7, 2:7] == 255).all() and (c[2, 22:, 18:] == 255).all()      # cutout rectangle (clipped at the border)
    np.testing.assert_array_equal(x1[0, :, 3:7, 2:7], x2[0, :, 3:7, 2:7])       # same erase values in A and B
    keep = np.ones((24, 20), bool); keep[3:7, 2:7] = False
    np.testing.assert_allclose(x1[0][:, keep], synth.normalize_nchw(a[:1])[0][:, keep], atol=2e-6)
    inside = (mask[0] >= 1) & keep
    np.testing.assert_allclose(x2[0][:, inside], synth.normalize_nchw(donor[:1])[0][:, inside], atol=2e-6)
    assert set(np.unique(c)) <= {0, 1, 255}
