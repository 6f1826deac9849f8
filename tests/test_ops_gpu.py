"""Per-op parity of the hand-written conv kernels through the C ABI (stcd_op_conv / stcd_op_wgrad).

Inputs and filters are bf16-representable, accumulation is fp32 in every implementation, so the MFMA kernel
(impl=1), the reference FMA kernel (impl=0) and the plain-C oracle (double accumulation) must agree to the rounding
of the bf16 OUTPUT (one ulp = 2^-8 relative) for conv, and to fp32 summation-order noise for the weight gradient."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ops_c as O
from stcd_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def geom(n, hi, wi, ci, ldi, hm, wm, in_stride, ho, wo, out_stride, oy0, ox0, co, ldo, taps):
    g = _lib.ConvGeom()
    g.n, g.hi, g.wi, g.ci, g.ldi = n, hi, wi, ci, ldi
    g.hm, g.wm, g.in_stride = hm, wm, in_stride
    g.ho, g.wo, g.out_stride, g.oy0, g.ox0 = ho, wo, out_stride, oy0, ox0
    g.co, g.ldo, g.ntaps = co, ldo, len(taps)
    for i, (dy, dx) in enumerate(taps):
        g.dy[i], g.dx[i] = dy, dx
    return g


TAPS3 = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]


def run_conv(impl, g, x, w, bias, out):
    l = _lib.lib()
    nbytes = l.stcd_op_scratch_bytes(C.byref(g))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(l.stcd_op_conv(_lib.DTYPE_BF16, impl, C.byref(g), C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()),
                              C.c_void_p(bias.data_ptr()) if bias is not None else None, C.c_void_p(out.data_ptr()),
                              C.c_void_p(scratch.data_ptr()), nbytes, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()


def run_wgrad(impl, g, x, dout):
    l = _lib.lib()
    nbytes = l.stcd_op_scratch_bytes(C.byref(g))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((g.ntaps, g.ci, g.co), float("nan"), dtype=torch.float32, device=DEV)
    _lib.check(l.stcd_op_wgrad(_lib.DTYPE_BF16, impl, C.byref(g), C.c_void_p(x.data_ptr()), C.c_void_p(dout.data_ptr()),
                               C.c_void_p(dw.data_ptr()), C.c_void_p(scratch.data_ptr()), nbytes,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return dw


def rnd(rng, *shape, scale=1.0):
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32)).bfloat16()


CONV_CASES = [  # n, h, w, ci, co   (ci includes the engine's zero padding to 8)
    (32, 64, 64, 16, 16), (3, 40, 48, 32, 32), (2, 64, 64, 8, 16), (2, 32, 48, 16, 32), (2, 32, 32, 32, 16),
    (2, 16, 16, 8, 16), (2, 16, 32, 16, 16), (1, 24, 16, 16, 32), (2, 16, 16, 32, 32), (1, 16, 16, 32, 64),
    (2, 8, 16, 64, 64), (1, 16, 16, 64, 128), (1, 8, 16, 128, 128), (1, 8, 16, 256, 128), (1, 8, 16, 384, 128),
    (1, 16, 16, 48, 16), (1, 16, 16, 96, 32), (1, 16, 16, 192, 64), (2, 16, 16, 16, 2), (1, 16, 16, 16, 1),
    (1, 20, 18, 16, 16), (1, 9, 11, 64, 64), (1, 25, 7, 32, 32),
    (16, 64, 64, 32, 32), (6, 48, 80, 64, 32), (40, 32, 32, 128, 64),     # persistent multi-tile walks of k_conv_res
]


@pytest.mark.parametrize("n,h,w,ci,co", CONV_CASES)
def test_conv3x3_mfma_vs_ref_vs_oracle(n, h, w, ci, co):
    rng = np.random.default_rng(ci * 1000 + co + h)
    x = rnd(rng, n, h, w, ci).to(DEV)
    wt = rnd(rng, 9, ci, co, scale=1.0 / np.sqrt(9 * ci)).float().to(DEV)
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(DEV)
    ldo = (co + 7) // 8 * 8
    g = geom(n, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, ldo, TAPS3)
    outs = []
    for impl in (0, 1, 2):
        out = torch.zeros(n, h, w, ldo, dtype=torch.bfloat16, device=DEV)
        run_conv(impl, g, x, wt, bias, out)
        outs.append(out[..., :co].float().cpu())
    # oracle: NCHW, weight [co][ci][ky][kx] with tap (dy,dx) -> (ky,kx) = (dy+1, dx+1)
    w_ref = wt.cpu().numpy().reshape(3, 3, ci, co).transpose(3, 2, 0, 1)
    ref = O.conv2d_fwd(x.float().cpu().numpy().transpose(0, 3, 1, 2), w_ref, bias.cpu().numpy(), 1).transpose(0, 2, 3, 1)
    for name, o in (("ref-kernel", outs[0]), ("mfma-auto", outs[1]), ("mfma-generic", outs[2])):
        np.testing.assert_allclose(o.numpy(), ref, rtol=2 ** -7, atol=2e-3, err_msg=name)
    np.testing.assert_allclose(outs[1].numpy(), outs[0].numpy(), rtol=2 ** -7, atol=2e-3)


HALO_CASES = [(1, 16, 16, 128, 128), (1, 16, 16, 128, 256), (2, 20, 37, 128, 256), (1, 33, 16, 256, 128), (3, 48, 48, 256, 256),
              (2, 8, 8, 128, 384), (1, 40, 24, 128, 512), (5, 128, 128, 128, 128), (9, 64, 64, 128, 256), (1, 24, 24, 384, 128)]


@pytest.mark.parametrize("n,h,w,ci,co", HALO_CASES)
def test_conv3x3_halo_kernel_vs_ref_vs_oracle(n, h, w, ci, co):
    """k_conv_halo (impl 3: resident 18 x 18 x 64-channel halo, filter streamed per (chunk, tap) stage through a 3-deep LDS ring,
    persistent 4-wave blocks of 16 x 16 pixels x 128 channels): ragged maps (masked border tiles), 2 ... 6 channel chunks, 1 ... 4
    output-channel slices, and more tiles than blocks (5 x 128 x 128: 320 tiles over 256 blocks; 9 x 64 x 64 x 256 channels: 144
    tiles over the 128 blocks of each slice -- the halo stream crosses tile boundaries)."""
    rng = np.random.default_rng(ci * 1000 + co + h)
    x = rnd(rng, n, h, w, ci).to(DEV)
    wt = rnd(rng, 9, ci, co, scale=1.0 / np.sqrt(9 * ci)).float().to(DEV)
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(DEV)
    g = geom(n, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, co, TAPS3)
    outs = []
    for impl in (0, 3):
        out = torch.zeros(n, h, w, co, dtype=torch.bfloat16, device=DEV)
        run_conv(impl, g, x, wt, bias, out)
        outs.append(out.float().cpu())
    w_ref = wt.cpu().numpy().reshape(3, 3, ci, co).transpose(3, 2, 0, 1)
    ref = O.conv2d_fwd(x.float().cpu().numpy().transpose(0, 3, 1, 2), w_ref, bias.cpu().numpy(), 1).transpose(0, 2, 3, 1)
    np.testing.assert_allclose(outs[1].numpy(), ref, rtol=2 ** -7, atol=2e-3, err_msg="halo kernel vs oracle")
    np.testing.assert_allclose(outs[1].numpy(), outs[0].numpy(), rtol=2 ** -7, atol=2e-3, err_msg="halo kernel vs reference kernel")


WGRAD_CASES = [(2, 40, 24, 192, 96), (1, 16, 16, 320, 32), (2, 16, 16, 8, 16), (2, 16, 32, 16, 16), (1, 24, 16, 16, 32), (2, 16, 16, 32, 32), (2, 8, 16, 64, 64),
               (1, 16, 16, 64, 128), (1, 8, 16, 128, 128), (1, 8, 16, 256, 128), (1, 16, 16, 48, 16), (1, 16, 16, 96, 32),
               (2, 16, 16, 16, 2), (1, 20, 18, 16, 16), (1, 9, 11, 64, 64), (4, 32, 32, 16, 16)]


@pytest.mark.parametrize("n,h,w,ci,co", WGRAD_CASES)
def test_wgrad_mfma_vs_ref_vs_oracle(n, h, w, ci, co):
    rng = np.random.default_rng(7 * ci + co + w)
    x = rnd(rng, n, h, w, ci).to(DEV)
    ldo = (co + 7) // 8 * 8
    dout = torch.zeros(n, h, w, ldo, dtype=torch.bfloat16, device=DEV)
    dout[..., :co] = rnd(rng, n, h, w, co).to(DEV)
    g = geom(n, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, ldo, TAPS3)
    dw0, dw1 = run_wgrad(0, g, x, dout).cpu().numpy(), run_wgrad(1, g, x, dout).cpu().numpy()
    _, dw_ref, _ = O.conv2d_bwd(x.float().cpu().numpy().transpose(0, 3, 1, 2), np.zeros((co, ci, 3, 3), np.float32),
                                dout[..., :co].float().cpu().numpy().transpose(0, 3, 1, 2), 1)
    ref = dw_ref.transpose(2, 3, 1, 0).reshape(9, ci, co)       # [co][ci][ky][kx] -> [tap][ci][co]
    scale = np.abs(ref).max()
    np.testing.assert_allclose(dw0 / scale, ref / scale, atol=2e-4, err_msg="ref-kernel")
    np.testing.assert_allclose(dw1 / scale, ref / scale, atol=2e-4, err_msg="mfma")
    if ci >= 64:          # the 64 x 32-channel tile (two 32-channel LDS sub-images, no k-split across waves)
        np.testing.assert_allclose(run_wgrad(4, g, x, dout).cpu().numpy() / scale, ref / scale, atol=2e-4, err_msg="mfma, wide tile")
    if ci >= 64 and co >= 64:      # the 64 x 64-channel tile (dY as two 32-channel sub-images too; opt-in: measured slower)
        np.testing.assert_allclose(run_wgrad(5, g, x, dout).cpu().numpy() / scale, ref / scale, atol=2e-4, err_msg="mfma, 64 x 64 tile")


# the LDS-DMA weight-gradient kernel (k_wgrad_dma, impl=7: block = (position split, tap, 256 x 256 channel tile), K-tiles of 64
# positions staged by `buffer_load ... lds`, transposed fragment reads): against an fp64 einsum of the same bf16 operands -- map
# widths that do and do not divide 64 (the per-row (y, x) carry), ragged split ends (positions not a multiple of 64: zero rows),
# several images (rows at the image seam must not see the neighbour image through a +-1 tap), 2 x 1 / 1 x 2 channel tiles, wider
# pixel pitches than channel counts, and the 4-tap lists of the transposed-conv phases.
WDMA_CASES = [  # n, h, w, ci, co, ldi, ldo, taps, out_stride, (oy0, ox0)
    (1, 64, 64, 256, 256, 256, 256, TAPS3, 1, (0, 0)), (2, 33, 67, 256, 256, 256, 256, TAPS3, 1, (0, 0)), (3, 16, 100, 256, 256, 320, 264, TAPS3, 1, (0, 0)),
    (1, 40, 64, 512, 256, 512, 256, TAPS3, 1, (0, 0)), (2, 24, 72, 256, 512, 256, 512, TAPS3, 1, (0, 0)),
    (2, 48, 96, 256, 256, 256, 256, [(0, 0), (0, 1), (1, 0), (1, 1)], 1, (0, 0)), (4, 128, 128, 256, 256, 256, 256, TAPS3, 1, (0, 0)),
    # the sub-pixel phases of a stride-2 transposed convolution: dY is the (oy0, ox0) phase of a map twice as large
    (2, 40, 72, 256, 256, 256, 256, [(0, 0), (0, 1), (1, 0), (1, 1)], 2, (1, 1)), (3, 21, 64, 256, 256, 256, 272, [(0, 0), (0, -1), (-1, 0), (-1, -1)], 2, (0, 1)),
    (1, 64, 64, 256, 256, 256, 256, [(0, 0), (1, 0)], 2, (1, 0)),
]


@pytest.mark.parametrize("n,h,w,ci,co,ldi,ldo,taps,so,o0", WDMA_CASES)
def test_lds_dma_wgrad_kernel(n, h, w, ci, co, ldi, ldo, taps, so, o0):
    rng = np.random.default_rng(ci + co + h + w)
    x = torch.zeros(n, h, w, ldi, dtype=torch.bfloat16, device=DEV)
    x[..., :ci] = rnd(rng, n, h, w, ci).to(DEV)
    x[..., ci:] = 3.0                                          # channels beyond Ci belong to another tensor: must not be read
    dfull = torch.zeros(n, h * so, w * so, ldo, dtype=torch.bfloat16, device=DEV)
    dfull[..., :co] = rnd(rng, n, h * so, w * so, co).to(DEV)  # the other phases hold data too: they must not be read
    dfull[..., co:] = 5.0
    g = geom(n, h, w, ci, ldi, h, w, 1, h * so, w * so, so, o0[0], o0[1], co, ldo, taps)
    dw = run_wgrad(7, g, x, dfull).double()
    dout = dfull[:, o0[0]::so, o0[1]::so]
    xd, dd = x[..., :ci].double(), dout[..., :co].double()
    want = torch.zeros(len(taps), ci, co, dtype=torch.float64, device=DEV)
    for t, (dy, dx) in enumerate(taps):                        # dW[t] = sum X(y + dy, x + dx)^T dY(y, x) over the positions whose tap is inside
        ys, xs = torch.arange(h, device=DEV) + dy, torch.arange(w, device=DEV) + dx
        oky, okx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
        sub = xd[:, ys.clamp(0, h - 1)][:, :, xs.clamp(0, w - 1)] * (oky.view(1, -1, 1, 1) & okx.view(1, 1, -1, 1))
        want[t] = sub.reshape(-1, ci).T @ dd.reshape(-1, co)
    scale = want.abs().max().item()
    assert (dw - want).abs().max().item() / scale <= 2e-5      # fp32 accumulation over <= 65536 positions per slab, then a slab sum
    if taps is TAPS3 and n * h * w <= 16384:
        ref = run_wgrad(0, geom(n, h, w, ci, ldi, h, w, 1, h, w, 1, 0, 0, co, ldo, taps), x, dfull).double()
        assert (dw - ref).abs().max().item() / scale <= 2e-4


def test_lds_dma_kernels_are_repeatable():
    """Race screen of the two LDS-DMA pipelines (counted vmcnt + barrier placement decide what a fragment read sees, and an early
    read passes whenever the DMA happens to land first): the kernels are deterministic, so 30 launches each at ChangeFormer's
    sizes (4096 / 1024 tiles over 256 persistent blocks; 28 position splits x 9 taps), with other traffic in between, must
    reproduce the first launch bit for bit."""
    rng = np.random.default_rng(5)
    for (n, h, w) in ((4, 512, 512), (4, 256, 256)):
        ci = co = 256
        x = rnd(rng, n, h, w, ci).to(DEV)
        wt = rnd(rng, 9, ci, co, scale=1.0 / np.sqrt(9 * ci)).float().to(DEV)
        dout = rnd(rng, n, h, w, co).to(DEV)
        g = geom(n, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, co, TAPS3)
        first_y = first_dw = None
        junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
        for it in range(30):
            out = torch.empty(n, h, w, co, dtype=torch.bfloat16, device=DEV)
            run_conv(6, g, x, wt, None, out)
            dw = run_wgrad(7, g, x, dout)
            if it % 3 == 0:
                junk.fill_(it)                                 # shifts cache state and timing between launches
            if first_y is None:
                first_y, first_dw = out.clone(), dw.clone()
            else:
                assert torch.equal(out, first_y), f"conv run {it} differs at {n}x{h}x{w}"
                assert torch.equal(dw, first_dw), f"wgrad run {it} differs at {n}x{h}x{w}"


def up_phase_taps(py, px):
    return [(dy, dx) for dy in range(py + 1) for dx in range(px + 1)]


@pytest.mark.parametrize("c", [16, 32, 64, 128])
def test_upconv_phases_and_stride2_dgrad(c):
    """ConvTranspose2d(k3,s2,p1,op1) as 4 sub-pixel phase launches writing a channel slice of a wider concat buffer,
    and its data gradient as a stride-2 conv: MFMA vs reference kernel (the reference kernel's whole-model use is
    pinned to the reference vectors in test_engine_gpu.py)."""
    rng = np.random.default_rng(c)
    n, h, w, ldcat = 2, 8, 16, 2 * c
    x = rnd(rng, n, h, w, c).to(DEV)
    outs = []
    for impl in (0, 1, 2):
        cat = torch.zeros(n, 2 * h, 2 * w, ldcat, dtype=torch.bfloat16, device=DEV)
        for py in (0, 1):
            for px in (0, 1):
                taps = up_phase_taps(py, px)
                wt = rnd(np.random.default_rng(c + 10 * py + px), len(taps), c, c, scale=0.1).float().to(DEV)
                g = geom(n, h, w, c, c, h, w, 1, 2 * h, 2 * w, 2, py, px, c, ldcat, taps)
                run_conv(impl, g, x, wt, None, cat)
        assert cat[..., c:].abs().max().item() == 0.0           # the neighbouring slice is untouched
        outs.append(cat[..., :c].float().cpu().numpy())
    np.testing.assert_allclose(outs[1], outs[0], rtol=2 ** -7, atol=2e-3)
    np.testing.assert_allclose(outs[2], outs[0], rtol=2 ** -7, atol=2e-3)
    # stride-2 3x3 conv over a [2h,2w] gradient held in a wider buffer
    dy_ = torch.zeros(n, 2 * h, 2 * w, ldcat, dtype=torch.bfloat16, device=DEV)
    dy_[..., :c] = rnd(rng, n, 2 * h, 2 * w, c).to(DEV)
    wt = rnd(rng, 9, c, c, scale=0.05).float().to(DEV)
    g = geom(n, 2 * h, 2 * w, c, ldcat, h, w, 2, h, w, 1, 0, 0, c, c, TAPS3)
    res = []
    for impl in (0, 1):
        o = torch.zeros(n, h, w, c, dtype=torch.bfloat16, device=DEV)
        run_conv(impl, g, dy_, wt, None, o)
        res.append(o.float().cpu().numpy())
    np.testing.assert_allclose(res[1], res[0], rtol=2 ** -7, atol=2e-3)
    # weight gradient of one phase (tap offsets on the input side, stride-2 offsets on the dY side)
    taps = up_phase_taps(1, 1)
    g = geom(n, h, w, c, c, h, w, 1, 2 * h, 2 * w, 2, 1, 1, c, ldcat, taps)
    d0, d1 = run_wgrad(0, g, x, dy_).cpu().numpy(), run_wgrad(1, g, x, dy_).cpu().numpy()
    np.testing.assert_allclose(d1 / np.abs(d0).max(), d0 / np.abs(d0).max(), atol=2e-4)


@pytest.mark.parametrize("n,h,w,ci,co", [(32, 256, 256, 16, 16), (32, 128, 128, 32, 32), (32, 64, 64, 64, 64), (32, 32, 32, 128, 128),
                                         (16, 32, 32, 256, 128),
                                         # SiamUnet_conc's widened first decoder conv of every stage (SiamUnet_conc.py:54,66,78,87)
                                         (16, 32, 32, 384, 128), (16, 64, 64, 192, 64), (16, 128, 128, 96, 32), (16, 256, 256, 48, 16),
                                         # SNUNet's dense-concat inputs (SNUNet.py:127-142): conv0_4, conv1_3, conv2_2, conv3_1
                                         (16, 256, 256, 224, 32), (16, 128, 128, 384, 64), (16, 64, 64, 640, 128), (16, 32, 32, 1024, 256)])
def test_full_size_conv_properties(n, h, w, ci, co):
    """The bench-sized layers of the engine (too large for the CPU oracle) through properties of a convolution that
    hold bit for bit, because every output position accumulates its (tap, channel) products in the same order
    wherever its tile sits: (a) translation equivariance -- shifting the input by (3, 5) pixels shifts the interior of
    the output; (b) batch independence -- image k of the batch equals the same image convolved alone; and
    (c) agreement with the plain-FMA reference kernel to the rounding of the bf16 output on a sampled image."""
    rng = np.random.default_rng(h + ci)
    x = rnd(rng, n, h, w, ci).to(DEV)
    wt = rnd(rng, 9, ci, co, scale=1.0 / np.sqrt(9 * ci)).float().to(DEV)
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(DEV)
    g = geom(n, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, co, TAPS3)
    out = torch.zeros(n, h, w, co, dtype=torch.bfloat16, device=DEV)
    run_conv(1, g, x, wt, bias, out)
    # (a) shift by (3, 5)
    sy, sx = 3, 5
    xs = torch.zeros_like(x)
    xs[:, sy:, sx:, :] = x[:, :h - sy, :w - sx, :]
    outs = torch.zeros_like(out)
    run_conv(1, g, xs, wt, bias, outs)
    assert torch.equal(outs[:, sy + 1:h - 1, sx + 1:w - 1, :], out[:, 1:h - 1 - sy, 1:w - 1 - sx, :])
    # (b) one image alone
    k = n // 2
    g1 = geom(1, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, co, TAPS3)
    o1 = torch.zeros(1, h, w, co, dtype=torch.bfloat16, device=DEV)
    run_conv(1, g1, x[k:k + 1].contiguous(), wt, bias, o1)
    assert torch.equal(o1[0], out[k])
    # (c) reference kernel on that image
    o0 = torch.zeros(1, h, w, co, dtype=torch.bfloat16, device=DEV)
    run_conv(0, g1, x[k:k + 1].contiguous(), wt, bias, o0)
    np.testing.assert_allclose(o1.float().cpu().numpy(), o0.float().cpu().numpy(), rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("n,h,w,ci,co,impl", [(32, 256, 256, 16, 16, 1), (32, 128, 128, 32, 32, 1), (32, 32, 32, 128, 128, 1), (16, 256, 256, 32, 16, 1),
                                               (32, 32, 32, 128, 128, 4), (16, 128, 128, 192, 64, 4)])
def test_full_size_wgrad_properties(n, h, w, ci, co, impl):
    """Bench-sized weight gradients: (a) exactly linear in dY for a power-of-two factor (fp32 slabs, fixed fold order);
    (b) invariant under a permutation of the batch up to fp32 summation order; (c) the sum over a batch equals the sum of
    the two half-batches' gradients (the K split over blocks must not lose or double-count positions)."""
    rng = np.random.default_rng(h * 3 + ci)
    x = rnd(rng, n, h, w, ci).to(DEV)
    dout = rnd(rng, n, h, w, co).to(DEV)
    g = geom(n, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, co, TAPS3)
    dw = run_wgrad(impl, g, x, dout)
    assert torch.isfinite(dw).all()
    dw2 = run_wgrad(impl, g, x, (dout.float() * 2).bfloat16())
    assert torch.equal(dw2, 2 * dw)
    perm = torch.from_numpy(rng.permutation(n)).to(DEV)
    dwp = run_wgrad(impl, g, x[perm].contiguous(), dout[perm].contiguous())
    scale = dw.abs().max().item()
    assert (dwp - dw).abs().max().item() < 2e-5 * scale
    gh = geom(n // 2, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, co, TAPS3)
    halves = run_wgrad(impl, gh, x[:n // 2].contiguous(), dout[:n // 2].contiguous()) + \
             run_wgrad(impl, gh, x[n // 2:].contiguous(), dout[n // 2:].contiguous())
    assert (halves - dw).abs().max().item() < 2e-5 * scale


# ---------------------------------------------------------------------------------------------------------------------
# Transposed / point-wise convolutions against the reference's per-op vectors (G1) and the plain-C oracle -- every
# implementation (reference FMA kernel in fp32 and bf16, MFMA auto-selected, MFMA generic), not MFMA vs FMA.
def _nhwc(x, dtype, ld=None):
    n, c, h, w = x.shape
    ld = ld or (c + 7) // 8 * 8
    t_ = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    t_[..., :c] = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(DEV).to(dtype)
    return t_


def _nchw(t_, c):
    return t_[..., :c].float().cpu().numpy().transpose(0, 3, 1, 2)


def _run_conv_dt(dt, impl, g, x, w, bias, out):
    l = _lib.lib()
    nbytes = l.stcd_op_scratch_bytes(C.byref(g))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(l.stcd_op_conv(dt, impl, C.byref(g), C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()),
                              C.c_void_p(bias.data_ptr()) if bias is not None else None, C.c_void_p(out.data_ptr()),
                              C.c_void_p(scratch.data_ptr()), nbytes, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()


def _run_wgrad_dt(dt, impl, g, x, dout):
    l = _lib.lib()
    nbytes = l.stcd_op_scratch_bytes(C.byref(g))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((g.ntaps, g.ci, g.co), float("nan"), dtype=torch.float32, device=DEV)
    _lib.check(l.stcd_op_wgrad(dt, impl, C.byref(g), C.c_void_p(x.data_ptr()), C.c_void_p(dout.data_ptr()), C.c_void_p(dw.data_ptr()),
                               C.c_void_p(scratch.data_ptr()), nbytes, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return dw.cpu().numpy()


IMPLS = [("fp32", 0), ("bf16", 0), ("bf16", 1), ("bf16", 2)]


def _q(a, dt):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy() if dt == "bf16" else np.asarray(a, np.float32)


def _tol(dt):
    return dict(rtol=2e-5, atol=2e-5) if dt == "fp32" else dict(rtol=2 ** -7, atol=4e-3)


def _pad8(c):
    return (c + 7) // 8 * 8


@pytest.mark.parametrize("dt,impl", IMPLS)
@pytest.mark.parametrize("tag,k", [("convT_s2_16_16", 3), ("convT_k2s2_32", 2)])
def test_stride2_transposed_conv_against_reference_vectors(golden, tag, k, dt, impl):
    """nn.ConvTranspose2d(k3,s2,p1,op1) (SiamUnet_diff.py:85) and ConvTranspose2d(k2,s2) (SNUNet.py:38): the four sub-pixel
    phase launches (forward, writing stride-2 positions of the output), the data gradient as a stride-2 gather conv and the
    per-phase weight gradients -- against the vectors captured from torch (fp32) / the oracle on bf16-rounded operands."""
    g = golden("g1_ops.npz")
    x, wt, b, gy = _q(g[f"{tag}/x"], dt), _q(g[f"{tag}/weight"], dt), g[f"{tag}/bias"], _q(g[f"{tag}/gy"], dt)
    n, ci, h, w = x.shape
    co = wt.shape[1]
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    dcode = _lib.DTYPE_F32 if dt == "fp32" else _lib.DTYPE_BF16
    s, p, op = (2, 1, 1) if k == 3 else (2, 0, 0)
    if dt == "fp32":
        y_ref, dx_ref, dw_ref = g[f"{tag}/y"], g[f"{tag}/dx"], g[f"{tag}/dweight"]
    else:
        y_ref = O.convT2d_fwd(x, wt, b, s, p, op)
        dx_ref, dw_ref, _ = O.convT2d_bwd(x, wt, gy, s, p, op)
    X = _nhwc(x, tdt)
    Y = torch.zeros(n, 2 * h, 2 * w, _pad8(co), dtype=tdt, device=DEV)
    dY = _nhwc(gy, tdt)
    bias = torch.from_numpy(b).to(DEV)
    dw_got = np.zeros_like(dw_ref)
    for py in (0, 1):
        for px in (0, 1):
            if k == 3:      # out(2m+py, 2n+px) = sum_{dy<=py, dx<=px} in(m+dy, n+dx) W[ci][co][py+1-2dy][px+1-2dx]
                taps = [(dy, dx) for dy in range(py + 1) for dx in range(px + 1)]
                kk = [(py + 1 - 2 * dy, px + 1 - 2 * dx) for dy, dx in taps]
            else:           # out(2m+py, 2n+px) = in(m, n) W[ci][co][py][px]
                taps, kk = [(0, 0)], [(py, px)]
            wp = torch.from_numpy(np.stack([wt[:, :, ky, kx] for ky, kx in kk]).astype(np.float32)).contiguous().to(DEV)   # [tap][ci][co]
            geo = geom(n, h, w, ci, _pad8(ci), h, w, 1, 2 * h, 2 * w, 2, py, px, co, _pad8(co), taps)
            _run_conv_dt(dcode, impl, geo, X, wp, bias, Y)
            if impl != 2:
                dwp = _run_wgrad_dt(dcode, min(impl, 1), geo, X, dY)
                for ti, (ky, kx) in enumerate(kk):
                    dw_got[:, :, ky, kx] = dwp[ti]
    np.testing.assert_allclose(_nchw(Y, co), y_ref, **_tol(dt))
    if impl != 2:
        sc = np.abs(dw_ref).max()
        np.testing.assert_allclose(dw_got / sc, dw_ref / sc, atol=2e-5 if dt == "fp32" else 2e-4)
    # data gradient: dIn(m, n)[ci] = sum_taps dOut(2m + dy, 2n + dx)[co] W[ci][co][ky][kx]
    if k == 3:
        taps = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
        kk = [(ky, kx) for ky in range(3) for kx in range(3)]
    else:
        taps = kk = [(dy, dx) for dy in range(2) for dx in range(2)]
    wd = torch.from_numpy(np.stack([wt[:, :, ky, kx].T for ky, kx in kk]).astype(np.float32)).contiguous().to(DEV)         # [tap][co][ci]
    geo = geom(n, 2 * h, 2 * w, co, _pad8(co), h, w, 2, h, w, 1, 0, 0, ci, _pad8(ci), taps)
    dX = torch.zeros(n, h, w, _pad8(ci), dtype=tdt, device=DEV)
    _run_conv_dt(dcode, impl, geo, dY, wd, None, dX)
    np.testing.assert_allclose(_nchw(dX, ci), dx_ref, **_tol(dt))


@pytest.mark.parametrize("dt,impl", IMPLS)
def test_pointwise_conv_against_reference_vectors(golden, dt, impl):
    """nn.Conv2d(128, 2, kernel_size=1) (SNUNet.py:106 conv_final): forward, data gradient (2 -> 128 channels through the
    zero-padded 8-channel gradient map) and weight gradient."""
    g = golden("g1_ops.npz")
    tag = "conv1x1_128_2"
    x, wt, b, gy = _q(g[f"{tag}/x"], dt), _q(g[f"{tag}/weight"], dt), g[f"{tag}/bias"], _q(g[f"{tag}/gy"], dt)
    n, ci, h, w = x.shape
    co = wt.shape[0]
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    dcode = _lib.DTYPE_F32 if dt == "fp32" else _lib.DTYPE_BF16
    if dt == "fp32":
        y_ref, dx_ref, dw_ref = g[f"{tag}/y"], g[f"{tag}/dx"], g[f"{tag}/dweight"]
    else:
        y_ref = O.conv2d_fwd(x, wt, b, 0)
        dx_ref, dw_ref, _ = O.conv2d_bwd(x, wt, gy, 0)
    X = _nhwc(x, tdt)
    Y = torch.zeros(n, h, w, 8, dtype=tdt, device=DEV)
    wp = torch.from_numpy(wt[:, :, 0, 0].T.astype(np.float32)[None]).contiguous().to(DEV)            # [1][ci][co]
    geo = geom(n, h, w, ci, ci, h, w, 1, h, w, 1, 0, 0, co, 8, [(0, 0)])
    _run_conv_dt(dcode, impl, geo, X, wp, torch.from_numpy(b).to(DEV), Y)
    np.testing.assert_allclose(_nchw(Y, co), y_ref, **_tol(dt))
    dY = _nhwc(gy, tdt)
    if impl != 2:
        dw = _run_wgrad_dt(dcode, min(impl, 1), geo, X, dY)                                             # [1][ci][co]
        sc = np.abs(dw_ref).max()
        np.testing.assert_allclose(dw[0].T[:, :, None, None] / sc, dw_ref / sc, atol=2e-5 if dt == "fp32" else 2e-4)
    wd = torch.zeros(1, 8, ci, dtype=torch.float32, device=DEV)                                        # [1][co padded to 8][ci]
    wd[0, :co] = torch.from_numpy(wt[:, :, 0, 0].astype(np.float32)).to(DEV)
    geo = geom(n, h, w, 8, 8, h, w, 1, h, w, 1, 0, 0, ci, ci, [(0, 0)])
    dX = torch.zeros(n, h, w, ci, dtype=tdt, device=DEV)
    _run_conv_dt(dcode, impl, geo, dY, wd, None, dX)
    np.testing.assert_allclose(_nchw(dX, ci), dx_ref, **_tol(dt))


# ---------------------------------------------------------------------------------------------------------------------
# 1x1 convolutions as the tiled GEMM kernel (k_conv_gemm, impl=1 picks it when Ci % 64 == 0 and Co % 64 == 0): SegCD's
# bottleneck / down-sample shapes, both tile sizes, ragged position counts, stride-2 gather (down-sample forward) and
# stride-2 scatter (its data gradient), against an fp64 matmul of the same bf16 operands and against the generic kernel.
GEMM_CASES = [  # n, h, w, ci, co, in_stride, out_stride
    (4, 32, 32, 64, 64, 1, 1), (2, 32, 32, 256, 64, 1, 1), (8, 32, 32, 64, 256, 1, 1), (2, 16, 16, 512, 128, 1, 1),
    (2, 8, 8, 2048, 512, 1, 1), (2, 8, 8, 512, 2048, 1, 1), (32, 16, 16, 256, 1024, 1, 1), (3, 6, 4, 128, 192, 1, 1),
    (1, 5, 7, 64, 128, 1, 1), (4, 32, 32, 256, 512, 2, 1), (3, 12, 20, 64, 128, 2, 1), (4, 16, 16, 512, 256, 1, 2),
]


@pytest.mark.parametrize("n,h,w,ci,co,si,so", GEMM_CASES)
def test_pointwise_gemm_kernel(n, h, w, ci, co, si, so):
    rng = np.random.default_rng(ci + 3 * co + h)
    hm, wm = h // si, w // si                       # positions computed
    ho, wo = hm * so, wm * so
    x = rnd(rng, n, h, w, ci).to(DEV)
    wt = rnd(rng, 1, ci, co, scale=1.0 / np.sqrt(ci)).float().to(DEV)
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(DEV)
    g = geom(n, h, w, ci, ci, hm, wm, si, ho, wo, so, 0, 0, co, co, [(0, 0)])
    outs = {}
    for impl in (1, 2):
        out = torch.full((n, ho, wo, co), 7.0, dtype=torch.bfloat16, device=DEV)
        run_conv(impl, g, x, wt, bias, out)
        outs[impl] = out.float()
    xs = x[:, ::si, ::si].double()[:, :hm, :wm]
    want = (xs.reshape(-1, ci) @ wt[0].double() + bias.double()).reshape(n, hm, wm, co)
    for impl in (1, 2):
        got = outs[impl][:, ::so, ::so]
        err = (got.double() - want).abs().max().item()
        assert err <= 2.0 ** -7 * want.abs().max().item(), (impl, err)          # one bf16 rounding of the output
        if so == 2:                                                            # positions the scatter must not touch
            untouched = outs[impl].clone()
            untouched[:, ::2, ::2] = 7.0
            assert torch.equal(untouched, torch.full_like(untouched, 7.0))
    assert (outs[1] - outs[2]).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()


# the same kernel walking a tap list: 3x3 layers whose filter slice exceeds LDS (impl=1 falls through the resident-filter kernel
# to it), stride-2 3x3 gathers, and the 2x2-tap stride-2 gather that is the data gradient of ConvTranspose2d(k=2, s=2)
TAPGEMM_CASES = [  # n, h, w, ci, co, stride, taps
    (2, 16, 16, 768, 128, 1, TAPS3), (1, 8, 8, 3072, 256, 1, TAPS3), (2, 9, 11, 512, 64, 1, TAPS3),
    (2, 32, 32, 128, 128, 2, TAPS3), (3, 12, 20, 64, 192, 2, TAPS3), (2, 16, 16, 256, 64, 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
]


@pytest.mark.parametrize("n,h,w,ci,co,stride,taps", TAPGEMM_CASES)
def test_tap_list_gemm_kernel(n, h, w, ci, co, stride, taps):
    rng = np.random.default_rng(ci + co + h + len(taps))
    hm, wm = h // stride, w // stride
    x = rnd(rng, n, h, w, ci).to(DEV)
    wt = rnd(rng, len(taps), ci, co, scale=1.0 / np.sqrt(len(taps) * ci)).float().to(DEV)
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(DEV)
    g = geom(n, h, w, ci, ci, hm, wm, stride, hm, wm, 1, 0, 0, co, co, taps)
    outs = {}
    for impl in (1, 2):
        out = torch.zeros(n, hm, wm, co, dtype=torch.bfloat16, device=DEV)
        run_conv(impl, g, x, wt, bias, out)
        outs[impl] = out.double()
    want = bias.double().view(1, 1, 1, co).expand(n, hm, wm, co).clone()
    xd = x.double()
    for t, (dy, dx) in enumerate(taps):                       # out(y, x) += X(y*s + dy, x*s + dx) . W[t]   (zero outside the image)
        ys = torch.arange(hm, device=DEV) * stride + dy
        xs = torch.arange(wm, device=DEV) * stride + dx
        oky, okx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
        sub = xd[:, ys.clamp(0, h - 1)][:, :, xs.clamp(0, w - 1)] * (oky.view(1, -1, 1, 1) & okx.view(1, 1, -1, 1))
        want += (sub.reshape(-1, ci) @ wt[t].double()).reshape(n, hm, wm, co)
    tol = 2.0 ** -7 * want.abs().max().item()
    for impl in (1, 2):
        assert (outs[impl] - want).abs().max().item() <= tol, impl


# the LDS-DMA kernel (k_conv_dma, impl=6: 256 positions x 256 channels per 8-wave block, (chunk, tap) K-tiles staged by
# `buffer_load ... lds` with counted vmcnt, two wave rows one barrier apart): ChangeFormer's 256 -> 256 3x3 layers and the
# 4-tap phases of its 4x4 stride-2 transposed convolutions (scatter at out_stride 2), ragged position counts (partial last
# tile: masked rows read zeros and are not stored), 1 ... 3 channel tiles, 4 ... 54 K-tiles, more tiles than CUs, and maps one
# tile row wide / narrower than a tile (taps of one tile reach three image rows or several images).  Every border tap of every
# case checks the zero fill of an out-of-range DMA.
DMA_CASES = [  # n, h, w, ci, co, out_stride, (oy0, ox0), taps
    (1, 16, 16, 128, 256, 1, (0, 0), TAPS3), (2, 20, 37, 128, 256, 1, (0, 0), TAPS3), (1, 33, 16, 256, 512, 1, (0, 0), TAPS3),
    (3, 48, 48, 256, 256, 1, (0, 0), TAPS3), (5, 128, 128, 128, 256, 1, (0, 0), TAPS3), (2, 64, 64, 384, 768, 1, (0, 0), TAPS3),
    (2, 24, 40, 256, 256, 2, (1, 0), [(0, 0), (0, 1), (1, 0), (1, 1)]), (2, 24, 40, 256, 256, 2, (0, 1), [(0, 0), (0, -1), (-1, 0), (-1, -1)]),
    (3, 17, 19, 256, 256, 1, (0, 0), [(0, 0)]), (1, 8, 8, 512, 256, 1, (0, 0), [(0, 0), (0, 1)]), (1, 300, 7, 64, 256, 1, (0, 0), TAPS3 + []),
]


@pytest.mark.parametrize("n,h,w,ci,co,so,o0,taps", DMA_CASES)
def test_lds_dma_conv_kernel(n, h, w, ci, co, so, o0, taps):
    if (ci // 64) * len(taps) % 2 or (ci // 64) * len(taps) < 4:
        pytest.skip("K-tile count must be even and >= 4")
    rng = np.random.default_rng(ci + co + h + len(taps))
    ho, wo = h * so, w * so
    x = rnd(rng, n, h, w, ci).to(DEV)
    wt = rnd(rng, len(taps), ci, co, scale=1.0 / np.sqrt(len(taps) * ci)).float().to(DEV)
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(DEV)
    g = geom(n, h, w, ci, ci, h, w, 1, ho, wo, so, o0[0], o0[1], co, co, taps)
    outs = {}
    for impl in (6, 2):
        out = torch.full((n, ho, wo, co), 7.0, dtype=torch.bfloat16, device=DEV)
        run_conv(impl, g, x, wt, bias, out)
        outs[impl] = out.double()
    want = bias.double().view(1, 1, 1, co).expand(n, h, w, co).clone()
    xd = x.double()
    for t, (dy, dx) in enumerate(taps):                       # out(y, x) += X(y + dy, x + dx) . W[t]   (zero outside the image)
        ys = torch.arange(h, device=DEV) + dy
        xs = torch.arange(w, device=DEV) + dx
        oky, okx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
        sub = xd[:, ys.clamp(0, h - 1)][:, :, xs.clamp(0, w - 1)] * (oky.view(1, -1, 1, 1) & okx.view(1, 1, -1, 1))
        want += (sub.reshape(-1, ci) @ wt[t].double()).reshape(n, h, w, co)
    tol = 2.0 ** -7 * want.abs().max().item()
    for impl in (6, 2):
        got = outs[impl][:, o0[0]::so, o0[1]::so]
        assert (got - want).abs().max().item() <= tol, impl
        if so == 2:                                           # positions the scatter must not touch
            untouched = outs[impl].clone()
            untouched[:, o0[0]::2, o0[1]::2] = 7.0
            assert torch.equal(untouched, torch.full_like(untouched, 7.0))
    # same operands, same fp32 accumulation order per output (chunk-major, tap, k): the two MFMA kernels agree to the last bit
    # wherever neither rounds differently -- allow one bf16 ulp
    assert (outs[6] - outs[2]).abs().max().item() <= tol


# one-tap weight gradients on the position-GEMM kernel (impl=3): 1x1 convs, stride-2 1x1 (in_stride 2), the phases of a 2x2
# stride-2 transposed conv (dY read at stride 2 from phase (py, px)), ragged position counts and channel counts that are not
# multiples of the 64 / 128 tile -- against an fp64 matmul of the same bf16 operands and the reference kernel (impl=0)
WGEMM_CASES = [  # n, h, w, ci, co, in_stride, out_stride, oy0, ox0
    (4, 32, 32, 64, 64, 1, 1, 0, 0), (2, 16, 16, 256, 128, 1, 1, 0, 0), (2, 8, 8, 2048, 512, 1, 1, 0, 0), (3, 5, 7, 192, 320, 1, 1, 0, 0),
    (8, 16, 16, 64, 256, 1, 1, 0, 0), (2, 16, 16, 128, 256, 2, 1, 0, 0), (2, 8, 8, 128, 128, 1, 2, 1, 0), (2, 8, 8, 64, 64, 1, 2, 1, 1),
    (32, 16, 16, 1024, 256, 1, 1, 0, 0),
]


@pytest.mark.parametrize("n,h,w,ci,co,si,so,oy0,ox0", WGEMM_CASES)
def test_weight_gradient_gemm_kernel(n, h, w, ci, co, si, so, oy0, ox0):
    rng = np.random.default_rng(ci + co + h + so)
    hm, wm = h // si, w // si
    ho, wo = hm * so, wm * so
    x = rnd(rng, n, h, w, ci).to(DEV)
    dout = rnd(rng, n, ho, wo, co).to(DEV)
    g = geom(n, h, w, ci, ci, hm, wm, si, ho, wo, so, oy0, ox0, co, co, [(0, 0)])
    got = run_wgrad(3, g, x, dout)[0].double()
    ref = run_wgrad(0, g, x, dout)[0].double()
    xs = x[:, ::si, ::si][:, :hm, :wm].double().reshape(-1, ci)
    ys = dout[:, oy0::so, ox0::so][:, :hm, :wm].double().reshape(-1, co)
    want = xs.T @ ys
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * scale + 1e-6       # fp32 accumulation order only
    assert (ref - want).abs().max().item() <= 2e-5 * scale + 1e-6


@pytest.mark.parametrize("n,h,w,ci,co,stride,taps", [(32, 64, 64, 64, 256, 1, [(0, 0)]), (32, 64, 64, 256, 64, 1, [(0, 0)]), (32, 8, 8, 2048, 512, 1, [(0, 0)]),
                                                     (32, 32, 32, 512, 1024, 2, [(0, 0)]), (32, 16, 16, 3072, 256, 1, TAPS3), (32, 64, 64, 128, 128, 2, TAPS3)])
def test_full_size_gemm_conv_properties(n, h, w, ci, co, stride, taps):
    """SegCD's bench-sized layers on the tap-list GEMM kernel (ResNet-50 bottleneck 1x1s, the stride-2 down-sample, the decoder's
    3072 -> 256 3x3, a stride-2 3x3): (a) batch independence, bit for bit -- image k of the 32-image batch equals the same image
    convolved alone (different tile walk, different position groups); (b) agreement with the plain-FMA reference kernel on that
    image to the rounding of the bf16 output."""
    rng = np.random.default_rng(h + ci + len(taps))
    hm, wm = h // stride, w // stride
    x = rnd(rng, n, h, w, ci).to(DEV)
    wt = rnd(rng, len(taps), ci, co, scale=1.0 / np.sqrt(len(taps) * ci)).float().to(DEV)
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(DEV)
    g = geom(n, h, w, ci, ci, hm, wm, stride, hm, wm, 1, 0, 0, co, co, taps)
    out = torch.zeros(n, hm, wm, co, dtype=torch.bfloat16, device=DEV)
    run_conv(1, g, x, wt, bias, out)
    assert torch.isfinite(out.float()).all()
    k = n // 2 + 1
    g1 = geom(1, h, w, ci, ci, hm, wm, stride, hm, wm, 1, 0, 0, co, co, taps)
    o1 = torch.zeros(1, hm, wm, co, dtype=torch.bfloat16, device=DEV)
    run_conv(1, g1, x[k:k + 1].contiguous(), wt, bias, o1)
    assert torch.equal(o1[0], out[k])
    o0 = torch.zeros(1, hm, wm, co, dtype=torch.bfloat16, device=DEV)
    run_conv(0, g1, x[k:k + 1].contiguous(), wt, bias, o0)
    np.testing.assert_allclose(o1.float().cpu().numpy(), o0.float().cpu().numpy(), rtol=2 ** -7, atol=2e-3)
