"""Per-op parity of the ChangeFormer kernels through the C ABI (stcd_op_cf_*), fp32 AND bf16, against torch's fp32 implementation of
the same op on the CPU (the tier's "plain PyTorch fp32 reference" for floating-point kernels), forward and backward:
im2col / col2im vs F.unfold / F.fold (i.e. OverlapPatchEmbed.proj and Attention.sr as GEMMs, ChangeFormer.py:207-208,315), LayerNorm
(:209,317,478,486), softmax attention with dropout on the probabilities (:347-354), depthwise 3x3 + GELU + dropout (:289-292,517-523),
residual + dropout + DropPath (:505-509), bilinear resize (:1585,1591), PReLU / dropout / ReLU / axpby (conv_diff :1138-1148,
ResidualBlock ChangeFormerBaseNetworks.py:109-120).  Dropout masks: the engine's counter hash, restated in
oracle/changeformer_ref.py (hash_keep / site_seed) -- the op must reproduce exactly those masks.

fp32: 2e-5 relative to the tensor's scale (sums 1e-4); bf16: inputs are rounded to bf16 first, the reference runs on the rounded
values in fp32, and the kernel must agree to the rounding of its bf16 OUTPUT."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import changeformer_ref as R
from stcd_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"fp32": (_lib.DTYPE_F32, torch.float32), "bf16": (_lib.DTYPE_BF16, torch.bfloat16)}
SEED = 0x5EED1234ABCD


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(x, dtype):
    """fp32 CPU tensor -> (device tensor in the activation dtype, the values it holds as fp32 CPU)"""
    d = x.to(DEV).to(DT[dtype][1]).contiguous()
    return d, d.float().cpu()


def f32(x):
    return x.float().to(DEV).contiguous()


def close(got, want, dtype, what, scale=None, fp32_tol=2e-5, bf16_tol=2 ** -7):
    got, want = got.float().cpu(), want.float()
    s = float(want.abs().max()) if scale is None else scale
    tol = (fp32_tol if dtype == "fp32" else bf16_tol) * max(s, 1e-6)
    err = float((got - want).abs().max())
    assert err <= tol, f"{what} [{dtype}]: max abs error {err:.3e} > {tol:.3e} (scale {s:.3e})"


def keep_mask(shape, p, site=0, seed=SEED):
    n = int(np.prod(shape))
    k = R.hash_keep(n, R.site_seed(seed, site), p).reshape(shape)
    return torch.from_numpy(k.astype(np.float32)) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def scratch_bytes(nbytes):
    return torch.zeros(int(nbytes), dtype=torch.uint8, device=DEV)


L = None


def lib():
    global L
    if L is None:
        L = _lib.lib()
    return L


def test_site_seed_and_hash_restatement_match_the_library():
    for seed in (0, 1, SEED, 2 ** 63 + 12345):
        for site in (0, 1, 7, 123):
            assert lib().stcd_cf_site_seed(C.c_uint64(seed), site) == R.site_seed(seed, site)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,h,w,c,k,stride,pad", [(2, 16, 16, 3, 7, 4, 3), (2, 12, 20, 64, 7, 2, 3), (1, 8, 8, 320, 7, 2, 3),
                                                   (2, 16, 24, 64, 8, 8, 0), (3, 8, 8, 128, 2, 2, 0), (1, 6, 10, 16, 3, 2, 1)])
def test_im2col_and_col2im_vs_unfold_fold(dtype, n, h, w, c, k, stride, pad):
    g = torch.Generator().manual_seed(n * 1000 + c + k)
    x = torch.randn(n, c, h, w, generator=g)
    xd, xq = dev(x.permute(0, 2, 3, 1), dtype)                       # NHWC
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    K = c * k * k
    ldc = (K + 63) // 64 * 64
    col = torch.full((n * ho * wo, ldc), 7.0, dtype=DT[dtype][1], device=DEV)
    _lib.check(lib().stcd_op_cf_im2col(DT[dtype][0], P(xd), P(col), ldc, n, h, w, c, k, stride, pad, stream()))
    want = F.unfold(xq.permute(0, 3, 1, 2), k, padding=pad, stride=stride)          # [n, c*k*k, L]: the reference's own K order
    want = want.transpose(1, 2).reshape(n * ho * wo, K)
    assert torch.equal(col[:, :K].float().cpu(), want), "im2col is a pure gather: must be bit-exact"
    assert float(col[:, K:].float().abs().max()) == 0.0 if ldc > K else True
    # col2im == fold (sum of the patches' contributions)
    dc = torch.randn(n * ho * wo, ldc, generator=g)
    dcd, dcq = dev(dc, dtype)
    for acc in (0, 1):
        base = torch.randn(n, h, w, c, generator=g)
        dxd, dxq = dev(base, dtype)
        _lib.check(lib().stcd_op_cf_col2im(DT[dtype][0], P(dcd), ldc, P(dxd), n, h, w, c, k, stride, pad, acc, stream()))
        fold = F.fold(dcq[:, :K].reshape(n, ho * wo, K).transpose(1, 2), (h, w), k, padding=pad, stride=stride).permute(0, 2, 3, 1)
        close(dxd, fold + (dxq if acc else 0), dtype, f"col2im acc={acc}", fp32_tol=1e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,c,eps", [(37, 64, 1e-5), (300, 128, 1e-6), (65, 320, 1e-6), (16, 512, 1e-5), (9, 1024, 1e-6)])
def test_layernorm_forward_backward(dtype, rows, c, eps):
    g = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, c, generator=g) * 1.5 + 0.7
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    xd, xq = dev(x, dtype)
    y = torch.empty_like(xd)
    stats = torch.empty(rows, 2, device=DEV)
    gd, bd = f32(gamma), f32(beta)          # (held in variables: a temporary's memory would be re-used by the next allocation)
    _lib.check(lib().stcd_op_cf_layernorm(DT[dtype][0], P(xd), P(y), P(gd), P(bd), P(stats), rows, c, eps, stream()))
    xr = xq.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    want = F.layer_norm(xr, (c,), gr, br, eps)
    close(y, want.detach(), dtype, "layernorm")
    np.testing.assert_allclose(stats[:, 0].cpu().numpy(), xq.mean(1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(stats[:, 1].cpu().numpy(), (1.0 / torch.sqrt(xq.var(1, unbiased=False) + eps)).numpy(), rtol=1e-4)
    # backward: dy (+ dy2), optional residual add
    dy, dy2, add = torch.randn(rows, c, generator=g), torch.randn(rows, c, generator=g), torch.randn(rows, c, generator=g)
    for use2, use_add in ((False, False), (True, True)):
        dyd, dyq = dev(dy, dtype)
        d2d, d2q = dev(dy2, dtype)
        ad, aq = dev(add, dtype)
        dx = torch.empty_like(xd)
        dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        sc = scratch_bytes(lib().stcd_op_cf_scratch_bytes(rows, c, 0, 0, 0))
        _lib.check(lib().stcd_op_cf_layernorm_bwd(DT[dtype][0], P(dyd), P(d2d) if use2 else None, P(xd), P(stats), P(gd),
                                                  P(ad) if use_add else None, P(dx), P(dg), P(db), P(sc), rows, c, stream()))
        for t_ in (xr, gr, br):
            t_.grad = None
        up = dyq + (d2q if use2 else 0)
        F.layer_norm(xr, (c,), gr, br, eps).backward(up)
        close(dx, xr.grad + (aq if use_add else 0), dtype, "layernorm dx", fp32_tol=5e-5)
        close(dg, gr.grad, "fp32", "layernorm dgamma", fp32_tol=1e-4)
        close(db, br.grad, "fp32", "layernorm dbeta", fp32_tol=1e-4)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,c", [(1000, 64), (77, 320), (5000, 2048), (3, 8)])
def test_colsum(dtype, rows, c):
    x = torch.randn(rows, c, generator=torch.Generator().manual_seed(rows))
    xd, xq = dev(x, dtype)
    out = torch.empty(c, device=DEV)
    sc = scratch_bytes(lib().stcd_op_cf_scratch_bytes(rows, c, 0, 0, 0))
    _lib.check(lib().stcd_op_cf_colsum(DT[dtype][0], P(xd), rows, c, P(out), P(sc), stream()))
    close(out, xq.double().sum(0).float(), "fp32", "colsum", scale=float(xq.abs().sum(0).max()), fp32_tol=1e-5)


def _attn_ref(q, kv, heads, mask):
    """Attention.forward between the projections (ChangeFormer.py:336-354), fp32, explicit attn_drop mask [n, heads, N, Nkv]."""
    n, N, C_ = q.shape
    d = C_ // heads
    qh = q.reshape(n, N, heads, d).permute(0, 2, 1, 3)
    kvh = kv.reshape(n, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kvh[0], kvh[1]
    attn = (qh @ k.transpose(-2, -1)) * (d ** -0.5)
    attn = attn.softmax(dim=-1)
    lse = torch.logsumexp((qh @ k.transpose(-2, -1)) * (d ** -0.5), dim=-1)
    if mask is not None:
        attn = attn * mask
    return (attn @ v).transpose(1, 2).reshape(n, N, C_), lse


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,N,Nkv,heads,d,p", [(2, 256, 4, 1, 64, 0.1), (2, 64, 4, 2, 64, 0.0), (1, 100, 37, 4, 80, 0.1), (2, 70, 70, 8, 64, 0.1),
                                                (1, 300, 64, 2, 32, 0.25), (1, 130, 33, 1, 128, 0.1), (2, 1000, 256, 2, 64, 0.1),
                                                (1, 700, 200, 4, 80, 0.1), (1, 64, 300, 1, 64, 0.1)])
def test_attention_forward_backward(dtype, n, N, Nkv, heads, d, p):
    g = torch.Generator().manual_seed(N * 7 + Nkv)
    C_ = heads * d
    q, kv, do = torch.randn(n, N, C_, generator=g), torch.randn(n, Nkv, 2 * C_, generator=g), torch.randn(n, N, C_, generator=g)
    qd, qq = dev(q, dtype)
    kd, kq = dev(kv, dtype)
    dd, dq_ = dev(do, dtype)
    out = torch.empty_like(qd)
    lse = torch.empty(n, heads, N, device=DEV)
    _lib.check(lib().stcd_op_cf_attention(DT[dtype][0], P(qd), P(kd), P(out), P(lse), n, N, Nkv, heads, d, p, C.c_uint64(SEED), stream()))
    mask = keep_mask((n, heads, N, Nkv), p) if p > 0 else None
    qr, kr = qq.clone().requires_grad_(True), kq.clone().requires_grad_(True)
    want, wl = _attn_ref(qr, kr, heads, mask)
    close(out, want.detach(), dtype, "attention out", fp32_tol=2e-5)
    np.testing.assert_allclose(lse.cpu().numpy(), wl.detach().numpy(), rtol=2e-5, atol=2e-5)
    dqo, dkv = torch.empty_like(qd), torch.empty_like(kd)
    sc = scratch_bytes(lib().stcd_op_cf_attention_scratch_bytes(n, N, Nkv, heads, d))
    # the backward consumes the STORED output (bf16-rounded in bf16 mode), as the engine does
    _lib.check(lib().stcd_op_cf_attention_bwd(DT[dtype][0], P(qd), P(kd), P(out), P(dd), P(lse), P(dqo), P(dkv), P(sc), n, N, Nkv, heads, d, p,
                                              C.c_uint64(SEED), stream()))
    want.backward(dq_)
    close(dqo, qr.grad, dtype, "attention dq", fp32_tol=5e-5, bf16_tol=2 ** -6)
    close(dkv, kr.grad, dtype, "attention dkv", fp32_tol=5e-5, bf16_tol=2 ** -6)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,h,w,ch,p", [(2, 16, 16, 256, 0.1), (1, 5, 7, 64, 0.0), (2, 8, 8, 1280, 0.1), (1, 3, 3, 2048, 0.3), (3, 1, 9, 8, 0.1)])
def test_dwconv_gelu_dropout_forward_backward(dtype, n, h, w, ch, p):
    g = torch.Generator().manual_seed(h * 31 + ch)
    x = torch.randn(n, h, w, ch, generator=g)
    wt, b = torch.randn(ch, 1, 3, 3, generator=g) * 0.3, torch.randn(ch, generator=g) * 0.1
    xd, xq = dev(x, dtype)
    u, a = torch.empty_like(xd), torch.empty_like(xd)
    wd, bd = f32(wt), f32(b)
    _lib.check(lib().stcd_op_cf_dwgelu(DT[dtype][0], P(xd), P(u), P(a), P(wd), P(bd), n, h, w, ch, p, C.c_uint64(SEED), stream()))
    mask = keep_mask((n, h, w, ch), p) if p > 0 else torch.ones(n, h, w, ch)
    xr, wr, br = xq.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ur = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1, groups=ch).permute(0, 2, 3, 1)
    close(u, ur.detach(), dtype, "dwconv u")
    # the activation is computed from the STORED u (one rounding in bf16 mode)
    uq = u.float().cpu()
    close(a, F.gelu(uq) * mask, dtype, "gelu+dropout a")
    da = torch.randn(n, h, w, ch, generator=g)
    dad, daq = dev(da, dtype)
    dh = torch.empty_like(xd)
    dw, db = torch.empty(ch, 9, device=DEV), torch.empty(ch, device=DEV)
    sc = scratch_bytes(lib().stcd_op_cf_dwgelu_scratch_bytes(n, h, w, ch))
    _lib.check(lib().stcd_op_cf_dwgelu_bwd(DT[dtype][0], P(xd), P(u), P(dad), P(dh), P(wd), P(dw), P(db), P(sc), n, h, w, ch, p,
                                           C.c_uint64(SEED), stream()))
    # reference gradient with the same stored u: g = da * mask * gelu'(u); then the depthwise conv's own backward
    uu = uq.clone().requires_grad_(True)
    (F.gelu(uu) * mask).backward(daq)
    gq = uu.grad if dtype == "fp32" else uu.grad.to(torch.bfloat16).float()
    close(dad, uu.grad, dtype, "gate gradient")
    ur.backward(gq)
    close(dh, xr.grad, dtype, "dwconv dh", fp32_tol=5e-5)
    close(dw, wr.grad.reshape(ch, 9), "fp32", "dwconv dw", scale=float(wr.grad.abs().max()), fp32_tol=2e-4 if dtype == "fp32" else 2e-3)
    close(db, br.grad, "fp32", "dwconv db", scale=float(br.grad.abs().max()), fp32_tol=2e-4 if dtype == "fp32" else 2e-3)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,rows,c,p,pp", [(4, 64, 64, 0.1, 0.3), (2, 100, 320, 0.0, 0.0), (6, 9, 8, 0.1, 0.5)])
def test_residual_dropout_droppath(dtype, n, rows, c, p, pp):
    g = torch.Generator().manual_seed(rows + c)
    x, y = torch.randn(n, rows, c, generator=g), torch.randn(n, rows, c, generator=g)
    xd, xq = dev(x, dtype)
    yd, yq = dev(y, dtype)
    out = torch.empty_like(xd)
    _lib.check(lib().stcd_op_cf_resid_drop(DT[dtype][0], P(xd), P(yd), P(out), n, rows, c, p, pp, C.c_uint64(SEED), 0, stream()))
    m = keep_mask((n, rows, c), p, 0) if p > 0 else torch.ones(n, rows, c)
    path = keep_mask((n,), pp, 1).reshape(n, 1, 1) if pp > 0 else torch.ones(n, 1, 1)
    if pp > 0:
        assert 0 < int((path == 0).sum()) < n or n < 4       # the test shapes exercise both branches
    close(out, xq + yq * m * path, dtype, "residual + dropout + droppath")
    dy = torch.empty_like(xd)
    _lib.check(lib().stcd_op_cf_resid_drop(DT[dtype][0], None, P(yd), P(dy), n, rows, c, p, pp, C.c_uint64(SEED), 1, stream()))
    close(dy, yq * m * path, dtype, "its gradient map")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,h,w,H,W,c", [(2, 4, 4, 8, 8, 64), (1, 2, 2, 16, 16, 256), (2, 4, 6, 16, 24, 8), (1, 8, 8, 32, 32, 64), (1, 5, 3, 7, 11, 16)])
def test_bilinear_forward_backward(dtype, n, h, w, H, W, c):
    g = torch.Generator().manual_seed(h * W + c)
    x = torch.randn(n, h, w, c, generator=g)
    xd, xq = dev(x, dtype)
    for acc in (0, 1):
        base = torch.randn(n, H, W, c, generator=g)
        od, oq = dev(base, dtype)
        _lib.check(lib().stcd_op_cf_bilinear(DT[dtype][0], P(xd), P(od), n, h, w, H, W, c, acc, 0, stream()))
        xr = xq.clone().requires_grad_(True)
        want = F.interpolate(xr.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        close(od, want.detach() + (oq if acc else 0), dtype, f"bilinear acc={acc}")
        if H == 2 * h and W == 2 * w:        # the scale_factor=2 form of ChangeFormer.py:1591 is the same map
            w2 = F.interpolate(xq.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear").permute(0, 2, 3, 1)
            assert torch.equal(w2, want.detach())
        dy = torch.randn(n, H, W, c, generator=g)
        dyd, dyq = dev(dy, dtype)
        b2 = torch.randn(n, h, w, c, generator=g)
        dxd, dxq = dev(b2, dtype)
        _lib.check(lib().stcd_op_cf_bilinear(DT[dtype][0], P(dyd), P(dxd), n, h, w, H, W, c, acc, 1, stream()))
        want.backward(dyq)
        close(dxd, xr.grad + (dxq if acc else 0), dtype, f"bilinear gradient acc={acc}", fp32_tol=5e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_elementwise_family(dtype):
    g = torch.Generator().manual_seed(5)
    rows, c = 333, 64
    x, y = torch.randn(rows, c, generator=g), torch.randn(rows, c, generator=g)
    x[0, :8] = 0.0                                                    # PReLU / ReLU at exactly zero
    xd, xq = dev(x, dtype)
    yd, yq = dev(y, dtype)
    alpha = torch.tensor([0.23])
    ald = f32(alpha)
    out = torch.empty_like(xd)
    ew = lambda op, a=None, af=0.0, bf=0.0, p=0.0, yy=None: _lib.check(lib().stcd_op_cf_elementwise(
        DT[dtype][0], op, P(xd), P(yy), P(out), rows, c, P(a), af, bf, p, C.c_uint64(SEED), stream()))
    ew(0, a=ald)
    close(out, F.prelu(xq, alpha), dtype, "prelu")
    ew(1, p=0.6)
    close(out, xq * keep_mask((rows, c), 0.6), dtype, "dropout(0.6)")
    ew(2)
    close(out, F.relu(xq), dtype, "relu")
    ew(3, yy=yd)
    close(out, xq * (yq > 0), dtype, "relu gradient")
    ew(4, af=0.1, bf=1.0, yy=yd)
    close(out, 0.1 * xq + yq, dtype, "axpby")
    ew(4, af=0.1)
    close(out, 0.1 * xq, dtype, "scale")
    # PReLU gradient: dy and d(alpha)
    dz = torch.randn(rows, c, generator=g)
    dzd, dzq = dev(dz, dtype)
    dy, da = torch.empty_like(xd), torch.empty(1, device=DEV)
    sc = scratch_bytes(lib().stcd_op_cf_scratch_bytes(rows, c, 0, 0, 0))
    _lib.check(lib().stcd_op_cf_prelu_bwd(DT[dtype][0], P(dzd), P(xd), P(dy), P(ald), P(da), P(sc), rows, c, stream()))
    xr, ar = xq.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    F.prelu(xr, ar).backward(dzq)
    close(dy, xr.grad, dtype, "prelu dy")
    close(da, ar.grad, "fp32", "prelu dalpha", scale=float((dzq * xq).abs().sum()), fp32_tol=1e-5)
