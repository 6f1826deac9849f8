"""Shared helpers of the parity tests."""
import re

import numpy as np
import torch


def t(a):
    return torch.from_numpy(np.asarray(a))


def grad_summary(g):
    g = g.detach().flatten().double().cpu()
    n = g.numel()
    idx = (np.arange(24) * max(n // 24, 1)) % n
    first = g[:8].numpy() if n >= 8 else np.pad(g.numpy(), (0, 8 - n))
    return np.concatenate([[g.sum().item(), g.norm().item()], first, g[idx].numpy()])


def zero_grad_by_construction(name):
    """A conv bias that feeds ONLY a train-mode BatchNorm has an exactly-zero gradient (BN subtracts the
    batch mean); autograd reports rounding noise, the engine reports 0."""
    if re.fullmatch(r"conv\d\dd?\.bias", name):
        return name != "conv11d.bias"
    if re.fullmatch(r"cross_conc\d\.(diff|conv_res)\.0\.bias", name):      # SiamUnet_cross_conc: both convs of the block feed a BatchNorm
        return True
    return name.endswith(".conv2.bias")


# Whole-network gradients are only piecewise continuous: a max-pool arg-max, a ReLU gate or |a-b|'s sign sitting on a
# near-tie flips under fp32 summation-order noise.  The REFERENCE does this to itself: perturbing the G2 inputs by 1e-6
# (relative) changes its own gradients by 2.7e-3 (relative l2), 3e-6 by 1.7e-2 (measured with the oracle, which is
# pinned to the reference).  Whole-model gradient checks are therefore PER-TENSOR: relative l2 error and cosine against
# the reference's gradient ("gf/" entries: the whole tensor up to GF_FULL elements, a fixed random sample of GF_FULL
# elements above that), bounded by the measured flip sensitivity.  A wrong sign, a dropped term or an all-zero
# gradient fails both; the tight per-element checks are per-op (test_ops_gpu.py, test_ew_ops_gpu.py), where no
# discontinuity exists.
REL_L2_MAX = 2e-2
COS_MIN = 0.9995
GF_FULL = 4096


def gf_index(name, numel):
    """Indices of the elements of a gradient tensor that the fixtures hold (all of them for small tensors)."""
    if numel <= GF_FULL:
        return np.arange(numel)
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()) + numel)
    return np.sort(rng.choice(numel, GF_FULL, replace=False))


def rel_l2_cos(got, ref):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    nr, ng = np.linalg.norm(ref), np.linalg.norm(got)
    rel = np.linalg.norm(got - ref) / max(nr, 1e-30)
    cos = float(got @ ref) / max(nr * ng, 1e-60)
    return rel, cos


ACHIEVED = {}      # test id -> (worst rel-l2, worst cosine) over the tensors it checked; printed by conftest at exit


def check_grad(name, got, g, rel_max=REL_L2_MAX, cos_min=COS_MIN, prefix="", tag=None):
    """got: gradient tensor; g: golden dict with '<prefix>gs/<name>' summaries and '<prefix>gf/<name>' (sampled) tensors."""
    ref = g[prefix + "gs/" + name]
    got = got.detach().float().cpu()
    if zero_grad_by_construction(name):
        assert abs(ref[2:]).max() < 1e-5 and got.abs().max().item() < 1e-5, name
        return None
    key = prefix + "gf/" + name
    assert key in g, f"fixture holds no reference gradient for {name}"
    want = np.asarray(g[key]).ravel()
    sample = got.flatten().numpy()[gf_index(name, got.numel())]
    assert sample.shape == want.shape, (name, sample.shape, want.shape)
    rel, cos = rel_l2_cos(sample, want)
    assert rel <= rel_max and cos >= cos_min, f"{name}: relative l2 error {rel:.3e} (max {rel_max}), cosine {cos:.6f} (min {cos_min})"
    # the tensor's l2 norm over ALL elements (the sample cannot see a fault outside it)
    np.testing.assert_allclose(got.double().norm().item(), ref[1], rtol=max(2 * rel_max, 1e-3), err_msg=name + " (l2 norm)")
    if tag is not None:
        w = ACHIEVED.get(tag, (0.0, 1.0))
        ACHIEVED[tag] = (max(w[0], rel), min(w[1], cos))
    return rel, cos
