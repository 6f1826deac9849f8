"""Per-op parity of the normalisation / pooling / fusion kernels through the C ABI (stcd_op_bn_act, stcd_op_bn_act_bwd,
stcd_op_bn_act_pair, stcd_op_maxpool[_bwd], stcd_op_fuse[_bwd], stcd_op_rep_pad[_bwd], stcd_op_skip_bwd), in fp32 AND bf16,
against (a) the per-op vectors captured from the reference's torch.nn modules (tests/golden/g1_ops.npz: bn/*, pool/*,
pool_odd/*, fuse/*, rpad/* -- ties included) and (b) the plain-C oracle on random maps (odd sizes, two BatchNorm groups,
Dropout2d masks with zeros).  fp32: 2e-5 (sums 1e-4); bf16: the inputs are rounded to bf16 first, the oracle runs on the
rounded values, and the kernel must agree to the rounding of its bf16 OUTPUT (2^-7 relative)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ops_c as O
from stcd_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"fp32": (_lib.DTYPE_F32, torch.float32), "bf16": (_lib.DTYPE_BF16, torch.bfloat16)}
TOL = {"fp32": dict(rtol=2e-5, atol=2e-5), "bf16": dict(rtol=2 ** -7, atol=4e-3)}
SUM_TOL = {"fp32": dict(rtol=1e-4, atol=1e-4), "bf16": dict(rtol=2e-3, atol=2e-3)}


def cpad(c):
    p = 8
    while p < c:
        p *= 2
    return p


def rq(x, dtype):
    """values as the activation dtype stores them (fp32 numpy)"""
    return torch.from_numpy(np.asarray(x, np.float32)).to(DT[dtype][1]).float().numpy()


def nhwc(x, dtype, ld=None):
    """NCHW fp32 numpy -> NHWC device tensor [N,H,W,ld] (channels padded with zeros)"""
    n, c, h, w = x.shape
    ld = ld or cpad(c)
    t = torch.zeros(n, h, w, ld, dtype=DT[dtype][1], device=DEV)
    t[..., :c] = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(DEV).to(DT[dtype][1])
    return t


def nchw(t, c):
    return t[..., :c].float().cpu().numpy().transpose(0, 3, 1, 2)


def mg(n, h, w, c, groups=1):
    g = _lib.MapGeom()
    g.n, g.h, g.w, g.c, g.groups = n, h, w, c, groups
    return g


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def scratch(g):
    n = _lib.lib().stcd_op_ew_scratch_bytes(C.byref(g))
    return torch.zeros(n, dtype=torch.uint8, device=DEV), n


def f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def pad1(v, cp, fill=0.0):
    o = np.full(cp, fill, np.float32)
    o[:len(v)] = v
    return o


def run_bn_act(dtype, x, gamma, beta, rm, rv, mask, relu, training, pool, groups=1):
    """x NCHW -> (A, P or None, stat, rm', rv') through stcd_op_bn_act"""
    n, c, h, w = x.shape
    cp = cpad(c)
    g = mg(n, h, w, cp, groups)
    Y = nhwc(x, dtype)
    A = torch.zeros_like(Y)
    Pt = torch.zeros(n, h // 2, w // 2, cp, dtype=Y.dtype, device=DEV) if pool else None
    gm, bt = f32(pad1(gamma, cp, 1.0)), f32(pad1(beta, cp))
    rmt, rvt = f32(pad1(rm, cp)), f32(pad1(rv, cp, 1.0))
    mk = None
    if mask is not None:
        mm = np.ones((n, cp), np.float32)
        mm[:, :c] = mask
        mk = f32(mm)
    stat = torch.zeros(groups * 4 * cp, dtype=torch.float32, device=DEV)
    sc, nb = scratch(g)
    _lib.check(_lib.lib().stcd_op_bn_act(DT[dtype][0], C.byref(g), P(Y), cp, P(gm), P(bt), P(rmt), P(rvt), P(mk), int(relu), int(training),
                                         P(A), cp, P(Pt), cp, P(stat), P(sc), nb, stream()))
    torch.cuda.synchronize()
    return A, Pt, stat, rmt.cpu().numpy()[:c], rvt.cpu().numpy()[:c], Y, mk


def run_bn_bwd(dtype, gy, Y, stat, mk, relu, n, c, h, w, groups=1):
    cp = cpad(c)
    g = mg(n, h, w, cp, groups)
    dA = nhwc(gy, dtype)
    dY = torch.zeros_like(dA)
    dg, db = torch.zeros(cp, dtype=torch.float32, device=DEV), torch.zeros(cp, dtype=torch.float32, device=DEV)
    sc, nb = scratch(g)
    _lib.check(_lib.lib().stcd_op_bn_act_bwd(DT[dtype][0], C.byref(g), P(dA), cp, P(Y), cp, P(stat), P(mk), int(relu), P(dY), cp, P(dg), P(db),
                                             P(sc), nb, stream()))
    torch.cuda.synchronize()
    return nchw(dY, c), dg.cpu().numpy()[:c], db.cpu().numpy()[:c]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_batchnorm_against_reference_vectors(golden, dtype):
    """nn.BatchNorm2d(16) train step of G1 (SiamUnet_diff.py:19): output, running statistics, eval output, dx, dweight, dbias."""
    g = golden("g1_ops.npz")
    x, gy = g["bn/x"], g["bn/gy"]
    n, c, h, w = x.shape
    A, _, stat, rm1, rv1, Y, _ = run_bn_act(dtype, x, g["bn/weight"], g["bn/bias"], g["bn/rm0"], g["bn/rv0"], None, False, True, False)
    if dtype == "fp32":
        np.testing.assert_allclose(nchw(A, c), g["bn/y"], **TOL[dtype])
        np.testing.assert_allclose(rm1, g["bn/rm1"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rv1, g["bn/rv1"], rtol=1e-6, atol=1e-7)
        dx_ref, dg_ref, db_ref = g["bn/dx"], g["bn/dweight"], g["bn/dbias"]
    else:       # the same step on bf16-rounded inputs, restated by the oracle (pinned to the vectors above on CPU)
        xr, gyr = rq(x, dtype), rq(gy, dtype)
        y_ref, mean, invstd, rm_ref, rv_ref = O.bn_train_fwd(xr, g["bn/weight"], g["bn/bias"], g["bn/rm0"], g["bn/rv0"])
        np.testing.assert_allclose(nchw(A, c), y_ref, **TOL[dtype])
        np.testing.assert_allclose(rm1, rm_ref, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rv1, rv_ref, rtol=1e-5, atol=1e-6)
        dx_ref, dg_ref, db_ref = O.bn_train_bwd(xr, gyr, g["bn/weight"], mean, invstd)
    dx, dg, db = run_bn_bwd(dtype, gy, Y, stat, None, False, n, c, h, w)
    np.testing.assert_allclose(dx, dx_ref, **TOL[dtype])
    np.testing.assert_allclose(dg, dg_ref, **SUM_TOL[dtype])
    np.testing.assert_allclose(db, db_ref, **SUM_TOL[dtype])
    # eval mode with the updated running statistics (bn/y_eval)
    Ae, _, _, _, _, _, _ = run_bn_act(dtype, x, g["bn/weight"], g["bn/bias"], g["bn/rm1"], g["bn/rv1"], None, False, False, False)
    ref = g["bn/y_eval"] if dtype == "fp32" else O.bn_eval_fwd(rq(x, dtype), g["bn/weight"], g["bn/bias"], g["bn/rm1"], g["bn/rv1"])
    np.testing.assert_allclose(nchw(Ae, c), ref, **TOL[dtype])


def oracle_bn_act(x, gamma, beta, rm, rv, mask, relu, groups, dtype):
    """conv output -> BN(train, per group sequentially) -> ReLU -> Dropout2d mask -> activation dtype; also the 2x2 pool"""
    n = x.shape[0]
    npg = n // groups
    a = np.empty_like(x)
    means, invstds = [], []
    for gi in range(groups):
        sl = slice(gi * npg, (gi + 1) * npg)
        y, mean, invstd, rm, rv = O.bn_train_fwd(x[sl], gamma, beta, rm, rv)
        if relu:
            y = np.maximum(y, 0)
        if mask is not None:
            y = y * mask[sl][:, :, None, None]
        a[sl] = y
        means.append(mean); invstds.append(invstd)
    a = rq(a, dtype)
    p, _ = O.maxpool2_fwd(a)
    return a, p, means, invstds, rm, rv


def oracle_bn_act_bwd(x, ga, gamma, beta, mask, relu, groups, means, invstds):
    n = x.shape[0]
    npg = n // groups
    dx = np.empty_like(x)
    dg, db = np.zeros(x.shape[1], np.float64), np.zeros(x.shape[1], np.float64)
    for gi in range(groups):
        sl = slice(gi * npg, (gi + 1) * npg)
        z = (x[sl] - means[gi][None, :, None, None]) * invstds[gi][None, :, None, None] * gamma[None, :, None, None] + beta[None, :, None, None]
        dz = ga[sl].copy()
        if mask is not None:
            dz = dz * mask[sl][:, :, None, None]
        if relu:
            dz = dz * (z > 0)
        d, g_, b_ = O.bn_train_bwd(x[sl], dz.astype(np.float32), gamma, means[gi], invstds[gi])
        dx[sl] = d
        dg += g_; db += b_
    return dx, dg.astype(np.float32), db.astype(np.float32)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,c,h,w,groups", [(4, 16, 12, 10, 2), (2, 32, 7, 5, 1), (6, 8, 9, 16, 2), (2, 128, 4, 6, 1), (4, 64, 16, 16, 2), (2, 512, 6, 6, 2), (4, 2048, 4, 4, 2), (2, 256, 8, 12, 1),
                                              (8, 64, 32, 32, 2), (16, 64, 64, 64, 2), (6, 16, 40, 24, 1)])
def test_bn_relu_dropout_pool_vs_oracle(dtype, n, c, h, w, groups):
    """BatchNorm2d(train) + ReLU + Dropout2d mask + 2x2 max-pool and its backward: odd sizes, two groups (the shared encoder
    BN sees date 0 then date 1: running statistics updated twice), masks with zeros.  The three largest cases give the reductions
    multi-block grids (64 ... 2048 blocks)."""
    rng = np.random.default_rng(n * 100 + c + h)
    x = rq(1.5 * rng.standard_normal((n, c, h, w)) + 0.3, dtype)
    gamma, beta = (1 + 0.1 * rng.standard_normal(c)).astype(np.float32), (0.1 * rng.standard_normal(c)).astype(np.float32)
    rm, rv = (0.1 * rng.standard_normal(c)).astype(np.float32), (1 + 0.2 * np.abs(rng.standard_normal(c))).astype(np.float32)
    mask = ((rng.random((n, c)) >= 0.2) * 1.25).astype(np.float32)
    A, Pt, stat, rm1, rv1, Y, mk = run_bn_act(dtype, x, gamma, beta, rm, rv, mask, True, True, True, groups)
    a_ref, p_ref, means, invstds, rm_ref, rv_ref = oracle_bn_act(x, gamma, beta, rm, rv, mask, True, groups, dtype)
    np.testing.assert_allclose(nchw(A, c), a_ref, **TOL[dtype])
    # the pooled map must be the maximum of the kernel's OWN activation, bit for bit
    own = torch.nn.functional.max_pool2d(A[..., :c].float().permute(0, 3, 1, 2), 2, 2).cpu().numpy()
    np.testing.assert_array_equal(nchw(Pt, c), own)
    np.testing.assert_allclose(rm1, rm_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv1, rv_ref, rtol=1e-5, atol=1e-6)
    ga = rq(rng.standard_normal((n, c, h, w)), dtype)
    dx, dg, db = run_bn_bwd(dtype, ga, Y, stat, mk, True, n, c, h, w, groups)
    dx_ref, dg_ref, db_ref = oracle_bn_act_bwd(x, ga, gamma, beta, mask, True, groups, means, invstds)
    np.testing.assert_allclose(dx, dx_ref, **TOL[dtype])
    scale = max(1.0, float(np.abs(dg_ref).max()))
    np.testing.assert_allclose(dg / scale, dg_ref / scale, **SUM_TOL[dtype])
    np.testing.assert_allclose(db / scale, db_ref / scale, **SUM_TOL[dtype])


def test_batchnorm_large_mean_cancellation():
    """|mean| >> std (un-normalised 0..255 imagery): the variance is formed as E[x^2] - mean^2 from fp32 chunk partials
    summed in double; bound the cancellation error against the double-precision oracle."""
    rng = np.random.default_rng(5)
    n, c, h, w = 4, 16, 64, 64
    x = (200.0 + 3.0 * rng.standard_normal((n, c, h, w))).astype(np.float32)
    gamma, beta = np.ones(c, np.float32), np.zeros(c, np.float32)
    A, _, stat, rm1, rv1, _, _ = run_bn_act("fp32", x, gamma, beta, np.zeros(c, np.float32), np.ones(c, np.float32), None, False, True, False)
    y_ref, mean, invstd, _, rv_ref = O.bn_train_fwd(x, gamma, beta, np.zeros(c, np.float32), np.ones(c, np.float32))
    st = stat.cpu().numpy().reshape(4, -1)
    np.testing.assert_allclose(st[0][:c], mean, rtol=1e-6)
    np.testing.assert_allclose(st[1][:c], invstd, rtol=2e-3)          # var ~ 9 out of E[x^2] ~ 4e4: ~1e-3 relative in fp32 partials
    np.testing.assert_allclose(nchw(A, c), y_ref, rtol=0, atol=1e-2)
    np.testing.assert_allclose(rv1, rv_ref, rtol=2e-3)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["pool", "pool_odd"])
def test_maxpool_against_reference_vectors(golden, dtype, tag):
    """F.max_pool2d(2,2) and its gradient incl. exact ties (post-ReLU zeros, a constant window) and an odd (7x5) map."""
    g = golden("g1_ops.npz")
    x, gy = rq(g[f"{tag}/x"], dtype), rq(g[f"{tag}/gy"], dtype)
    n, c, h, w = x.shape
    cp = cpad(c)
    geo = mg(n, h, w, cp)
    A = nhwc(x, dtype)
    Pt = torch.zeros(n, h // 2, w // 2, cp, dtype=A.dtype, device=DEV)
    l = _lib.lib()
    _lib.check(l.stcd_op_maxpool(DT[dtype][0], C.byref(geo), P(A), cp, P(Pt), cp, stream()))
    y_ref = g[f"{tag}/y"] if dtype == "fp32" else O.maxpool2_fwd(x)[0]
    np.testing.assert_array_equal(nchw(Pt, c), y_ref)
    dP = nhwc(gy, dtype)
    dA = torch.full_like(A, float("nan"))
    _lib.check(l.stcd_op_maxpool_bwd(DT[dtype][0], C.byref(geo), P(A), cp, P(dP), cp, P(dA), cp, 0, stream()))
    dx_ref = g[f"{tag}/dx"] if dtype == "fp32" else O.maxpool2_bwd(x, gy)
    np.testing.assert_array_equal(nchw(dA, c), dx_ref)
    # accumulate form: dA += routed gradient
    base = rq(np.random.default_rng(1).standard_normal(x.shape), dtype)
    dA2 = nhwc(base, dtype)
    _lib.check(l.stcd_op_maxpool_bwd(DT[dtype][0], C.byref(geo), P(A), cp, P(dP), cp, P(dA2), cp, 1, stream()))
    np.testing.assert_allclose(nchw(dA2, c), rq(base + dx_ref, dtype), **TOL[dtype])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_skip_fusion_against_reference_vectors(golden, dtype):
    """|a - b| (SiamUnet_diff.py:150) with exact ties (abs' = 0 there) and b - a (SiamUnet_sub.py:150), forward and gradient."""
    g = golden("g1_ops.npz")
    a, b, gg = rq(g["fuse/a"], dtype), rq(g["fuse/b"], dtype), rq(g["fuse/g"], dtype)
    n, c, h, w = a.shape
    cp = cpad(c)
    geo = mg(2 * n, h, w, cp, 2)
    AB = nhwc(np.concatenate([a, b]), dtype)
    l = _lib.lib()
    for mode in (0, 1):
        D = torch.zeros(n, h, w, cp, dtype=AB.dtype, device=DEV)
        _lib.check(l.stcd_op_fuse(DT[dtype][0], mode, C.byref(geo), P(AB), cp, P(D), cp, stream()))
        ref = O.fuse_fwd(a, b, mode)
        if dtype == "fp32":
            np.testing.assert_array_equal(nchw(D, c), g["fuse/abs"] if mode == 0 else b - a)
        np.testing.assert_allclose(nchw(D, c), ref, **TOL[dtype])
        dD = nhwc(gg, dtype)
        dAB = torch.full_like(AB, float("nan"))
        _lib.check(l.stcd_op_fuse_bwd(DT[dtype][0], mode, C.byref(geo), P(AB), cp, P(dD), cp, P(dAB), cp, stream()))
        da_ref, db_ref = O.fuse_bwd(a, b, gg, mode)
        out = nchw(dAB, c)
        np.testing.assert_array_equal(out[:n], da_ref)
        np.testing.assert_array_equal(out[n:], db_ref)
        if dtype == "fp32" and mode == 0:
            np.testing.assert_array_equal(out[:n], g["fuse/abs_da"])
            np.testing.assert_array_equal(out[n:], g["fuse/abs_db"])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_replication_pad_against_reference_vectors(golden, dtype):
    """nn.ReplicationPad2d((0,1,0,1)) (SiamUnet_diff.py:149) in place on the padded buffer, and its gradient."""
    g = golden("g1_ops.npz")
    x, y, gy = rq(g["rpad/x"], dtype), g["rpad/y"], rq(g["rpad/gy"], dtype)
    n, c, h0, w0 = x.shape
    H, W = y.shape[2:]
    cp = cpad(c)
    geo = mg(n, H, W, cp)
    buf = np.zeros((n, c, H, W), np.float32)
    buf[:, :, :h0, :w0] = x
    D = nhwc(buf, dtype)
    l = _lib.lib()
    _lib.check(l.stcd_op_rep_pad(DT[dtype][0], C.byref(geo), P(D), cp, h0, w0, stream()))
    np.testing.assert_array_equal(nchw(D, c), y if dtype == "fp32" else O.rep_pad_fwd(x, H, W))
    dD = nhwc(gy, dtype)
    _lib.check(l.stcd_op_rep_pad_bwd(DT[dtype][0], C.byref(geo), P(dD), cp, h0, w0, stream()))
    ref = g["rpad/dx"] if dtype == "fp32" else O.rep_pad_bwd(gy, h0, w0)
    np.testing.assert_allclose(nchw(dD, c)[:, :, :h0, :w0], ref, **TOL[dtype])
    # wider pads (two rows / three columns), as the 36x44 and 100x100 maps produce
    rng = np.random.default_rng(3)
    x2 = rq(rng.standard_normal((2, 8, 5, 6)), dtype)
    buf = np.zeros((2, 8, 7, 9), np.float32)
    buf[:, :, :5, :6] = x2
    D = nhwc(buf, dtype)
    geo = mg(2, 7, 9, 8)
    _lib.check(l.stcd_op_rep_pad(DT[dtype][0], C.byref(geo), P(D), 8, 5, 6, stream()))
    np.testing.assert_array_equal(nchw(D, 8), O.rep_pad_fwd(x2, 7, 9))
    g2 = rq(rng.standard_normal((2, 8, 7, 9)), dtype)
    dD = nhwc(g2, dtype)
    _lib.check(l.stcd_op_rep_pad_bwd(DT[dtype][0], C.byref(geo), P(dD), 8, 5, 6, stream()))
    np.testing.assert_allclose(nchw(dD, 8)[:, :, :5, :6], O.rep_pad_bwd(g2, 5, 6), **TOL[dtype])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("b,c,h,w", [(2, 16, 12, 10), (1, 32, 7, 9), (3, 8, 16, 16), (2, 128, 4, 4), (2, 128, 32, 32), (1, 16, 96, 256),
                                     (1, 64, 33, 71)])
def test_encoder_skip_layer_forward_and_backward_vs_oracle(dtype, mode, b, c, h, w):
    """Last conv of an encoder level, both dates in one pass.  Forward (stcd_op_bn_act_pair): BN(train) per date + ReLU +
    Dropout2d + pool + the skip |a1-a2| / a2-a1.  Backward (stcd_op_skip_bwd): pool gradient (first maximum) + fusion
    gradient (sign, 0 at ties) + BN backward -- against the composition of the oracle's per-op functions."""
    rng = np.random.default_rng(b * 1000 + c * 10 + h + mode)
    n = 2 * b
    x = rq(rng.standard_normal((n, c, h, w)), dtype)
    x[b:, 0] = x[:b, 0]                               # date-1 channel 0 == date-0 channel 0: exact ties in the fusion ...
    gamma, beta = (1 + 0.1 * rng.standard_normal(c)).astype(np.float32), (0.1 * rng.standard_normal(c)).astype(np.float32)
    rm, rv = np.zeros(c, np.float32), np.ones(c, np.float32)
    mask = ((rng.random((n, c)) >= 0.2) * 1.25).astype(np.float32)
    mask[b:, 0] = mask[:b, 0]                         # ... provided both dates keep the channel
    cp = cpad(c)
    geo = mg(n, h, w, cp, 2)
    Y = nhwc(x, dtype)
    A = torch.zeros_like(Y)
    Pt = torch.zeros(n, h // 2, w // 2, cp, dtype=Y.dtype, device=DEV)
    F = torch.zeros(b, h, w, cp, dtype=Y.dtype, device=DEV)
    mm = np.ones((n, cp), np.float32); mm[:, :c] = mask
    mk = f32(mm)
    gm, bt, rmt, rvt = f32(pad1(gamma, cp, 1.0)), f32(pad1(beta, cp)), f32(pad1(rm, cp)), f32(pad1(rv, cp, 1.0))
    stat = torch.zeros(2 * 4 * cp, dtype=torch.float32, device=DEV)
    sc, nb = scratch(geo)
    l = _lib.lib()
    _lib.check(l.stcd_op_bn_act_pair(DT[dtype][0], C.byref(geo), P(Y), cp, P(gm), P(bt), P(rmt), P(rvt), P(mk), mode, P(A), cp, P(Pt), cp,
                                     P(F), cp, P(stat), P(sc), nb, stream()))
    torch.cuda.synchronize()
    a_ref, p_ref, means, invstds, rm_ref, rv_ref = oracle_bn_act(x, gamma, beta, rm, rv, mask, True, 2, dtype)
    np.testing.assert_allclose(nchw(A, c), a_ref, **TOL[dtype])
    a_own = nchw(A, c)                                # downstream checks use the kernel's own (rounded) activations
    np.testing.assert_array_equal(nchw(Pt, c), O.maxpool2_fwd(a_own)[0])
    np.testing.assert_allclose(nchw(F, c), rq(O.fuse_fwd(a_own[:b], a_own[b:], mode), dtype), **TOL[dtype])
    if dtype == "fp32" and mode == 0:
        assert np.abs(nchw(F, c)[:, 0]).max() == 0.0  # the tied channel
    np.testing.assert_allclose(rmt.cpu().numpy()[:c], rm_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvt.cpu().numpy()[:c], rv_ref, rtol=1e-5, atol=1e-6)

    gd = rq(rng.standard_normal((b, c, h, w)), dtype)                   # gradient of the decoder's concat slice
    gp = rq(rng.standard_normal((n, c, h // 2, w // 2)), dtype)         # gradient of the pooled map (both dates)
    dD, dP = nhwc(gd, dtype), nhwc(gp, dtype)
    dA = torch.full_like(A, float("nan"))
    dY = torch.zeros_like(A)
    dg, db = torch.zeros(cp, dtype=torch.float32, device=DEV), torch.zeros(cp, dtype=torch.float32, device=DEV)
    _lib.check(l.stcd_op_skip_bwd(DT[dtype][0], mode, C.byref(geo), P(A), cp, P(Y), cp, P(dD), cp, P(dP), cp, P(stat), P(mk), P(dA), cp,
                                  P(dY), cp, P(dg), P(db), P(sc), nb, stream()))
    torch.cuda.synchronize()
    da1, da2 = O.fuse_bwd(a_own[:b], a_own[b:], gd, mode)
    da_ref = rq(np.concatenate([da1, da2]) + O.maxpool2_bwd(a_own, gp), dtype)
    np.testing.assert_allclose(nchw(dA, c), da_ref, **TOL[dtype])
    dx_ref, dg_ref, db_ref = oracle_bn_act_bwd(x, nchw(dA, c), gamma, beta, mask, True, 2, means, invstds)
    np.testing.assert_allclose(nchw(dY, c), dx_ref, **TOL[dtype])
    scale = max(1.0, float(np.abs(dg_ref).max()))
    np.testing.assert_allclose(dg.cpu().numpy()[:c] / scale, dg_ref / scale, **SUM_TOL[dtype])
    np.testing.assert_allclose(db.cpu().numpy()[:c] / scale, db_ref / scale, **SUM_TOL[dtype])

    # the engine's default plan (round 4): the forward writes no activations (a == NULL), the backward recomputes them from y
    # (k_skip_bwd_pair, both dates in one thread) -- the same pooled map, fused skip and dA bit for bit, the same sums and dY up to the
    # summation order of the block partials
    if cp & (cp - 1) == 0:
        P2, F2 = torch.zeros_like(Pt), torch.zeros_like(F)
        stat2 = torch.zeros_like(stat)
        rm2, rv2 = f32(pad1(rm, cp)), f32(pad1(rv, cp, 1.0))
        _lib.check(l.stcd_op_bn_act_pair(DT[dtype][0], C.byref(geo), P(Y), cp, P(gm), P(bt), P(rm2), P(rv2), P(mk), mode, None, cp, P(P2), cp,
                                         P(F2), cp, P(stat2), P(sc), nb, stream()))
        torch.cuda.synchronize()
        assert torch.equal(P2, Pt) and torch.equal(F2, F) and torch.equal(stat2, stat)
        dA2 = torch.full_like(A, float("nan"))
        dY2 = torch.zeros_like(A)
        dg2, db2 = torch.zeros_like(dg), torch.zeros_like(db)
        _lib.check(l.stcd_op_skip_bwd(DT[dtype][0], mode, C.byref(geo), None, cp, P(Y), cp, P(dD), cp, P(dP), cp, P(stat), P(mk), P(dA2), cp,
                                      P(dY2), cp, P(dg2), P(db2), P(sc), nb, stream()))
        torch.cuda.synchronize()
        assert torch.equal(dA2, dA), float((dA2.float() - dA.float()).abs().max())
        # (float sums of the same per-thread partials over differently cut blocks)
        np.testing.assert_allclose(dg2.cpu().numpy(), dg.cpu().numpy(), rtol=3e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(db2.cpu().numpy(), db.cpu().numpy(), rtol=3e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(dY2.float().cpu().numpy(), dY.float().cpu().numpy(), rtol=1e-2 if dtype == "bf16" else 1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,c,h,w", [(2, 64, 32, 32), (3, 8, 6, 10), (1, 128, 16, 8)])
def test_stem_maxpool3_and_gradient_vs_torch(dtype, n, c, h, w):
    """F.max_pool2d(x, 3, 2, 1) (ResNet stem, models/resnet.py:176) and its gradient through the recorded winners, on ReLU-ed
    inputs (most windows hold several equal zeros: torch routes the gradient to the FIRST maximum in scan order) -- against
    torch's own max_pool2d / autograd on the same stored values."""
    rng = np.random.default_rng(n + c + h)
    x = np.maximum(rng.standard_normal((n, c, h, w)).astype(np.float32), 0.0)
    x[:, :, ::3, ::2] = np.maximum(x[:, :, ::3, ::2], 0.5)                 # and some equal positive values
    xq = rq(x, dtype)
    X = nhwc(xq, dtype)
    g = mg(n, h, w, cpad(c))
    ld = cpad(c)
    Pl = torch.zeros(n, h // 2, w // 2, ld, dtype=DT[dtype][1], device=DEV)
    idx = torch.zeros(n * (h // 2) * (w // 2) * ld, dtype=torch.uint8, device=DEV)
    l = _lib.lib()
    _lib.check(l.stcd_op_maxpool3(DT[dtype][0], C.byref(g), P(X), ld, P(Pl), ld, P(idx), stream()))
    xt = torch.from_numpy(xq).to(DEV).requires_grad_(True)
    pt = torch.nn.functional.max_pool2d(xt, 3, 2, 1)
    np.testing.assert_array_equal(nchw(Pl, c), pt.detach().cpu().numpy())
    gp = rq(rng.standard_normal(pt.shape).astype(np.float32), dtype)
    pt.backward(torch.from_numpy(gp).to(DEV))
    dA = torch.full((n, h, w, ld), 7.0, dtype=DT[dtype][1], device=DEV)
    _lib.check(l.stcd_op_maxpool3_bwd(DT[dtype][0], C.byref(g), P(idx), P(nhwc(gp, dtype)), ld, P(dA), ld, stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(nchw(dA, c), rq(xt.grad.cpu().numpy(), dtype), **SUM_TOL[dtype])   # up to 4 windows summed per pixel


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,c,h,w", [(2, 16, 12, 10), (1, 32, 7, 5), (3, 8, 16, 16), (2, 128, 8, 8), (2, 16, 64, 64), (1, 64, 33, 17)])
def test_pairwise_depthwise_conv_vs_torch(dtype, n, c, h, w):
    """cross_conc's Conv2d(2C, C, 3, padding=1, groups=C) on the channel-interleaved pair (SiamUnet_crossconc.py:14-18,24-29) as the
    engine runs it -- on the two dates stacked in the batch dimension, nothing interleaved in memory -- against torch's grouped
    convolution of the interleaved tensor: forward, both dates' data gradients, the filter gradient; odd maps, 1 ... 16 channel blocks,
    multi-chunk filter-gradient grids."""
    rng = np.random.default_rng(c * 100 + h + n)
    x1, x2 = rq(rng.standard_normal((n, c, h, w)), dtype), rq(rng.standard_normal((n, c, h, w)), dtype)
    wt = (rng.standard_normal((c, 2, 3, 3)) / 3).astype(np.float32)
    bias = (0.1 * rng.standard_normal(c)).astype(np.float32)
    gy = rq(rng.standard_normal((n, c, h, w)), dtype)
    # torch reference on the interleaved tensor (fp64: the fixed point both dtypes are compared with)
    t1, t2 = torch.from_numpy(x1).double().requires_grad_(True), torch.from_numpy(x2).double().requires_grad_(True)
    tw = torch.from_numpy(wt).double().requires_grad_(True)
    inter = torch.stack((t1, t2), 2).reshape(n, 2 * c, h, w)
    y = torch.nn.functional.conv2d(inter, tw, torch.from_numpy(bias).double(), padding=1, groups=c)
    y.backward(torch.from_numpy(gy).double())
    geo = mg(2 * n, h, w, c, 2)
    A = nhwc(np.concatenate([x1, x2]), dtype, ld=c)
    out = torch.zeros(n, h, w, c, dtype=A.dtype, device=DEV)
    l = _lib.lib()
    W_, B_ = f32(wt), f32(bias)
    _lib.check(l.stcd_op_pairdw(DT[dtype][0], C.byref(geo), P(A), c, P(W_), P(B_), P(out), c, stream()))
    np.testing.assert_allclose(nchw(out, c), y.detach().numpy(), **TOL[dtype])
    dO = nhwc(gy, dtype, ld=c)
    dA = torch.full_like(A, float("nan"))
    dw = torch.full((c, 2, 3, 3), float("nan"), dtype=torch.float32, device=DEV)
    nb = l.stcd_op_pairdw_scratch_bytes(C.byref(geo))
    sc = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    _lib.check(l.stcd_op_pairdw_bwd(DT[dtype][0], C.byref(geo), P(A), c, P(dO), c, P(W_), P(dA), c, P(dw), P(sc), nb, stream()))
    torch.cuda.synchronize()
    got = nchw(dA, c)
    np.testing.assert_allclose(got[:n], t1.grad.numpy(), **TOL[dtype])
    np.testing.assert_allclose(got[n:], t2.grad.numpy(), **TOL[dtype])
    scale = float(tw.grad.abs().max())
    np.testing.assert_allclose(dw.cpu().numpy() / scale, tw.grad.numpy() / scale, rtol=0, atol=2e-5 if dtype == "fp32" else 2e-5)
