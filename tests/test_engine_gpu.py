"""Parity of the HIP engine (through the nn.Module boundary -> C ABI) against the golden vectors captured
from the reference and against the CPU oracle.  fp32 mode carries the 1e-3 bar of BASELINE.json;
bf16 mode is checked against the same vectors at a bf16-sized tolerance (stated per test)."""
import numpy as np
import pytest
import torch

from oracle import fcsiam_ref as R
from stcd_amd.modules import SiamUnet_conc, SiamUnet_cross_conc, SiamUnet_diff, SiamUnet_sub, Unet
from tests._util import check_grad, rel_l2_cos, t

pytestmark = pytest.mark.gpu
CLS = {"diff": SiamUnet_diff, "conc": SiamUnet_conc, "sub": SiamUnet_sub, "fcef": Unet, "xconc": SiamUnet_cross_conc}
DEV = "cuda:0"


def unwrap(o):
    return o[-1] if isinstance(o, (list, tuple)) else o


def loss_fn(label, logits, tgt):
    if label == 2:
        return torch.nn.functional.cross_entropy(logits, tgt)
    p = torch.sigmoid(logits)
    y = tgt.float().unsqueeze(1)
    bce = torch.nn.functional.binary_cross_entropy(p, y)
    dice = 1 - (2 * (p * y).sum() + 1) / (p.sum() + y.sum() + 1)
    return bce + dice


@pytest.mark.parametrize("arch", ["diff", "conc", "sub", "fcef", "xconc"])
@pytest.mark.parametrize("label", [1, 2])
def test_fp32_matches_reference_vectors(golden, arch, label):
    g = golden(f"g2_{arch}_{label}.npz")
    seed = int(g["seed"])
    x1, x2 = t(g["x1"]).to(DEV), t(g["x2"]).to(DEV)
    m = CLS[arch](3, label, dtype="fp32")
    m.load_state_dict(R.synth_state(arch, 3, label, seed, perturb_running=True))
    m.to(DEV).eval()
    with torch.no_grad():
        out = unwrap(m(x1, x2)).cpu().numpy()
        np.testing.assert_allclose(out, g["logits_eval"], rtol=1e-3, atol=1e-4)
        out64 = unwrap(m(t(g["y1"]).to(DEV), t(g["y2"]).to(DEV))).cpu().numpy()
        np.testing.assert_allclose(out64, g["logits_eval_64"], rtol=1e-3, atol=1e-4)

    m = CLS[arch](3, label, dtype="fp32")
    m.load_state_dict(R.synth_state(arch, 3, label, seed))
    m.to(DEV).train()
    m.set_dropout_masks(R.synth_masks(arch, 2, seed + 3))
    logits = unwrap(m(x1, x2))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits_train"], rtol=1e-3, atol=1e-4)
    loss = loss_fn(label, logits, t(g["target"]).to(DEV))
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    for name, p in m.named_parameters():       # per-tensor relative l2 <= 2e-2 and cosine >= 0.9995 vs the reference's gradient
        check_grad(name, p.grad, g, tag=f"fp32 {arch}({label}) vs reference G2")
    sd = m.state_dict()
    for k in [k for k in g if k.startswith("rs/")]:
        np.testing.assert_allclose(sd[k[3:]].cpu().numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("arch", ["diff", "conc"])
def test_fp32_odd_size_replication_pad(golden, arch):
    g = golden("g6_odd.npz")
    seed = int(g[f"{arch}/seed"])
    rng = np.random.default_rng(seed + 1)
    a = rng.standard_normal((1, 3, 100, 100)).astype(np.float32)
    b = (a + 0.5 * rng.standard_normal((1, 3, 100, 100))).astype(np.float32)
    m = CLS[arch](3, 2, dtype="fp32")
    m.load_state_dict(R.synth_state(arch, 3, 2, seed, perturb_running=True))
    m.to(DEV).eval()
    with torch.no_grad():
        out = m(t(a).to(DEV), t(b).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(out, g[f"{arch}/logits"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("arch", ["diff", "conc", "sub", "fcef", "xconc"])
def test_fp32_odd_size_backward_matches_oracle(arch):
    """ReplicationPad2d branch in training (36x44 -> pool chain 18x22, 9x11, 4x5, 2x2: two padded levels)."""
    seed, label = 77, 2
    rng = np.random.default_rng(seed)
    x1 = t(rng.standard_normal((2, 3, 36, 44)).astype(np.float32))
    x2 = t(rng.standard_normal((2, 3, 36, 44)).astype(np.float32))
    tgt = t((rng.random((2, 36, 44)) < 0.3).astype(np.int64))
    masks = R.synth_masks(arch, 2, seed + 1)
    st = R.synth_state(arch, 3, label, seed)
    m = CLS[arch](3, label, dtype="fp32")
    m.load_state_dict(st)
    m.to(DEV).train()
    m.set_dropout_masks(masks)
    loss = torch.nn.functional.cross_entropy(unwrap(m(x1.to(DEV), x2.to(DEV))), tgt.to(DEV))
    loss.backward()
    ref = {k: v.clone() for k, v in st.items()}
    names = [k for k, v in ref.items() if v.dtype.is_floating_point and "running" not in k]
    for k in names:
        ref[k].requires_grad_(True)
    rl = R.cross_entropy(R.forward(arch, ref, x1, x2, training=True, masks=masks), tgt)
    rl.backward()
    assert abs(loss.item() - rl.item()) < 1e-4
    from tests._util import ACHIEVED, COS_MIN, REL_L2_MAX
    for name, p in m.named_parameters():
        r = ref[name].grad
        if r.abs().max().item() < 1e-5:
            assert p.grad.abs().max().item() < 1e-5, name
            continue
        rel, cos = rel_l2_cos(p.grad.cpu().numpy(), r.numpy())
        assert rel <= REL_L2_MAX and cos >= COS_MIN, f"{name}: relative l2 error {rel:.3e}, cosine {cos:.6f}"
        w = ACHIEVED.get(f"fp32 {arch} 36x44 vs oracle", (0.0, 1.0))
        ACHIEVED[f"fp32 {arch} 36x44 vs oracle"] = (max(w[0], rel), min(w[1], cos))


@pytest.mark.parametrize("arch", ["diff", "conc"])
def test_bf16_tracks_reference_vectors(golden, arch):
    """bf16 storage + fp32 accumulation through ~20 layers on a 32x32 input (the bottleneck BN normalises over
    only 2x2x2 samples, which amplifies rounding): mean |dlogit| < 4e-2, max < 0.3 (|logit| ~ 1), loss within 2e-2,
    gradient cosine vs the reference's gradient for EVERY tensor of >= 256 elements: > 0.97 for the decoder's levels 1-2,
    > 0.8 for everything whose gradient passes the 2x2 bottleneck.  The tight bf16 checks are per-op (test_ops_gpu.py,
    test_ew_ops_gpu.py); the production-shaped whole-model bound is the 128x128 test below."""
    label = 2
    g = golden(f"g2_{arch}_{label}.npz")
    seed = int(g["seed"])
    x1, x2 = t(g["x1"]).to(DEV), t(g["x2"]).to(DEV)
    m = CLS[arch](3, label, dtype="bf16")
    m.load_state_dict(R.synth_state(arch, 3, label, seed))
    m.to(DEV).train()
    m.set_dropout_masks(R.synth_masks(arch, 2, seed + 3))
    logits = m(x1, x2)
    err = np.abs(logits.detach().cpu().numpy() - g["logits_train"])
    assert err.mean() < 4e-2 and err.max() < 0.3, (err.mean(), err.max())
    loss = loss_fn(label, logits, t(g["target"]).to(DEV))
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    loss.backward()
    from tests._util import ACHIEVED, gf_index, zero_grad_by_construction
    worst = {"dec": 1.0, "enc": 1.0}
    for name, p in m.named_parameters():       # EVERY parameter tensor with >= 256 elements, against the reference's gradient
        if zero_grad_by_construction(name) or p.numel() < 256:
            continue
        a = p.grad.flatten().cpu().double().numpy()[gf_index(name, p.numel())]
        _, cos = rel_l2_cos(a, g["gf/" + name])
        mod = name.split(".")[0]
        level = int(mod[6]) if mod.startswith("upconv") else int(mod[4] if mod.startswith("conv") else mod[2])
        dec = (mod.endswith("d") or mod.startswith("upconv")) and level <= 2       # decoder levels 1-2
        worst["dec" if dec else "enc"] = min(worst["dec" if dec else "enc"], cos)
        # shallow decoder tight; everything that passes the 2x2 bottleneck (BatchNorm over 8 samples amplifies every bf16
        # rounding) loose -- the production-shaped bound is test_train_step_128_tracks_reference_vectors
        assert cos > (0.97 if dec else 0.8), (name, cos)
    ACHIEVED[f"bf16 {arch}(2) vs reference G2 [worst cosine: decoder, encoder]"] = (worst["dec"], worst["enc"])


def test_fp32_config1_step(golden):
    """BASELINE.json configs[0]: SiamUnet_diff 3-ch 256x256 pair, batch 2, one optimizer step -- loss and change
    mask against the reference's CPU run (1e-3), gradient norms, AdamW/Adam deltas, BN running stats."""
    from stcd_amd import synth

    g = golden("g3_cfg1.npz")
    a, b, lab = synth.make_batch(2, 256, 256, seed=int(g["data_seed"]))
    A, B, L = t(a).to(DEV), t(b).to(DEV), t(lab).to(DEV)
    for tag, label, opt_name in (("ce", 2, "adamw"), ("cd", 1, "adam")):
        seed = int(g[f"{tag}/seed"])
        m = SiamUnet_diff(3, label, dtype="fp32")
        m.load_state_dict(R.synth_state("diff", 3, label, seed))
        m.to(DEV).train()
        m.set_dropout_masks(R.synth_masks("diff", 2, seed + 3))
        if opt_name == "adamw":
            opt = torch.optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
        else:
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
        before = {k: v.detach().clone() for k, v in m.named_parameters()}
        opt.zero_grad()
        logits = m(A, B)
        loss = loss_fn(label, logits, L)
        loss.backward()
        assert abs(loss.item() - float(g[f"{tag}/loss"])) < 1e-3
        lf = logits.detach().flatten().cpu()
        np.testing.assert_allclose(lf[t(g[f"{tag}/logits_sample_idx"])].numpy(), g[f"{tag}/logits_sample"], rtol=1e-3, atol=1e-3)
        pred = (logits.argmax(1) if label == 2 else (torch.sigmoid(logits[:, 0]) > 0.5).long()).cpu().numpy().astype(np.uint8)
        ref_mask = np.unpackbits(g[f"{tag}/mask_packed"])[:pred.size].reshape(pred.shape)
        assert (pred != ref_mask).mean() < 1e-3, "change mask differs from the reference on more than 0.1% of pixels"
        for name, p in m.named_parameters():
            check_grad(name, p.grad, g, prefix=f"{tag}/", tag=f"fp32 diff config-1 step ({tag}) vs reference G3")
        opt.step()
        for k in ("conv11.weight", "bn33.weight", "conv12d.weight"):
            d = (dict(m.named_parameters())[k].detach() - before[k]).cpu().numpy()
            # Adam's first step is lr*sign-like (|delta| ~ 1e-3): compare where the reference moved clearly
            ref = g[f"{tag}/delta/{k}"]
            np.testing.assert_allclose(d, ref, atol=2.5e-4, err_msg=k)
        sd = m.state_dict()
        for k in ("bn11", "bn43", "bn12d"):
            np.testing.assert_allclose(sd[f"{k}.running_mean"].cpu().numpy(), g[f"{tag}/rs/{k}.running_mean"], rtol=1e-3, atol=1e-5)
            np.testing.assert_allclose(sd[f"{k}.running_var"].cpu().numpy(), g[f"{tag}/rs/{k}.running_var"], rtol=1e-3, atol=1e-5)


def test_fp32_five_step_trajectory(golden):
    from stcd_amd import synth

    g = golden("g4_traj.npz")
    a, b, lab = synth.make_batch(2, 64, 64, seed=int(g["data_seed"]))
    A, B, L = t(a).to(DEV), t(b).to(DEV), t(lab).to(DEV)
    seed = int(g["seed"])
    m = SiamUnet_diff(3, 2, dtype="fp32")
    m.load_state_dict(R.synth_state("diff", 3, 2, seed))
    m.to(DEV).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    losses = []
    for step in range(5):
        m.set_dropout_masks(R.synth_masks("diff", 2, seed + 10 + step))
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(A, B), L)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], atol=5e-3)


_TOGGLES = ("STCD_FORCE_REF_KERNELS", "STCD_NO_SMALL_KERNEL", "STCD_NO_RES_KERNEL", "STCD_NO_WGRAD_GROUPS", "STCD_NO_SKIP_FUSED",
            "STCD_NO_ACT_FUSE")


@pytest.mark.parametrize("arch,n,h,w,min_cos", [("diff", 4, 64, 64, 0.99), ("sub", 4, 64, 64, 0.99), ("conc", 4, 64, 64, 0.99),
                                                 ("diff", 4, 50, 38, 0.96), ("sub", 2, 50, 38, 0.96)])
def test_bf16_kernel_families_agree(arch, n, h, w, min_cos):
    """The same bf16 step through (a) the reference FMA kernels, (b) the generic MFMA kernels only (separate BN
    statistics, one weight-gradient launch per layer, unfused skip backward), (c) the default path (small-channel and
    resident-filter persistent kernels with fused statistics, grouped weight gradients, one-pass skip backward), also on
    an odd-sized map (ReplicationPad2d branch): identical math up to summation order; a 1-ulp bf16 flip early in the net
    is amplified downstream, so the check is mean |dlogit| < 2e-2, max < 0.2, loss within 1e-2, gradient cosine > 0.99
    (> 0.96 on the 50x38 maps, whose 3x2 bottleneck normalises over a dozen samples), running stats within 2e-3."""
    import os

    seed, label = 123, 2
    rng = np.random.default_rng(seed)
    x1 = t(rng.standard_normal((n, 3, h, w)).astype(np.float32)).to(DEV)
    x2 = t(rng.standard_normal((n, 3, h, w)).astype(np.float32)).to(DEV)
    tgt = t((rng.random((n, h, w)) < 0.3).astype(np.int64)).to(DEV)
    st = R.synth_state(arch, 3, label, seed)
    masks = R.synth_masks(arch, n, seed + 1)
    res = {}
    generic = {k: "1" for k in _TOGGLES[1:]}
    for tag, env in (("ref", {"STCD_FORCE_REF_KERNELS": "1"}), ("generic", generic), ("default", {})):
        for k in _TOGGLES:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            m = CLS[arch](3, label, dtype="bf16")
        finally:
            for k in env:
                os.environ.pop(k, None)
        m.load_state_dict(st)
        m.to(DEV).train()
        m.set_dropout_masks(masks)
        logits = m(x1, x2)
        logits = logits[-1] if isinstance(logits, list) else logits
        loss = torch.nn.functional.cross_entropy(logits, tgt)
        loss.backward()
        g = torch.cat([p.grad.flatten() for p in m.parameters()]).cpu().double()
        res[tag] = (logits.detach().cpu(), loss.item(), g, m.state_dict()["bn12.running_var"].cpu())
    for tag in ("generic", "default"):
        d = (res[tag][0] - res["ref"][0]).abs()
        assert d.mean().item() < 2e-2 and d.max().item() < 0.2, (tag, d.mean().item(), d.max().item())
        assert abs(res[tag][1] - res["ref"][1]) < 1e-2, tag
        cos = (res[tag][2] @ res["ref"][2] / (res[tag][2].norm() * res["ref"][2].norm())).item()
        assert cos > min_cos, (tag, cos)
        np.testing.assert_allclose(res[tag][3].numpy(), res["ref"][3].numpy(), rtol=2e-3)


@pytest.mark.parametrize("label", [1, 2])
def test_snunet_fp32_matches_reference_vectors(golden, label):
    """SNUNet_ECAM (SNUNet.py:63-152) through the engine in fp32: eval logits, train logits, loss, every parameter
    gradient and the BN running statistics against the vectors captured from the reference."""
    from oracle import snunet_ref as S
    from stcd_amd.modules import SNUNet_ECAM

    g = golden(f"g2_snunet_{label}.npz")
    seed = int(g["seed"])
    x1, x2 = t(g["x1"]).to(DEV), t(g["x2"]).to(DEV)
    m = SNUNet_ECAM(3, label, dtype="fp32")
    m.load_state_dict(S.synth_state(3, label, seed, perturb_running=True))
    m.to(DEV).eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x1, x2).cpu().numpy(), g["logits_eval"], rtol=1e-3, atol=2e-4)
    m = SNUNet_ECAM(3, label, dtype="fp32")
    m.load_state_dict(S.synth_state(3, label, seed))
    m.to(DEV).train()
    logits = m(x1, x2)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits_train"], rtol=1e-3, atol=2e-4)
    loss = loss_fn(label, logits, t(g["target"]).to(DEV))
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    for name, p in m.named_parameters():
        check_grad(name, p.grad, g, tag=f"fp32 snunet({label}) vs reference G2")
    sd = m.state_dict()
    for k in [k for k in g if k.startswith("rs/")]:
        np.testing.assert_allclose(sd[k[3:]].cpu().numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_snunet_bf16_tracks_fp32_engine():
    """bf16 MFMA path of SNUNet against the fp32 engine path on a 64x64 pair: mean |dlogit| < 3e-2, loss within 2e-2,
    gradient cosine > 0.95 overall."""
    from oracle import snunet_ref as S
    from stcd_amd.modules import SNUNet_ECAM

    seed, label = 77, 2
    rng = np.random.default_rng(seed)
    x1 = t(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)).to(DEV)
    x2 = t(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)).to(DEV)
    tgt = t((rng.random((2, 64, 64)) < 0.3).astype(np.int64)).to(DEV)
    st = S.synth_state(3, label, seed)
    res = {}
    for dtype in ("fp32", "bf16"):
        m = SNUNet_ECAM(3, label, dtype=dtype)
        m.load_state_dict(st)
        m.to(DEV).train()
        logits = m(x1, x2)
        loss = torch.nn.functional.cross_entropy(logits, tgt)
        loss.backward()
        res[dtype] = (logits.detach().cpu(), loss.item(), torch.cat([p.grad.flatten() for p in m.parameters()]).cpu().double())
    d = (res["bf16"][0] - res["fp32"][0]).abs()
    assert d.mean().item() < 3e-2, d.mean().item()
    assert abs(res["bf16"][1] - res["fp32"][1]) < 2e-2
    cos = (res["bf16"][2] @ res["fp32"][2] / (res["bf16"][2].norm() * res["fp32"][2].norm())).item()
    assert cos > 0.95, cos


def test_full_size_properties_bf16():
    """BASELINE.json's bench configuration (SiamUnet_diff, 256x256, 16 pairs, bf16) through size-independent properties:
    (a) eval mode treats pairs independently: the 16-pair batch equals its two 8-pair halves bit for bit;
    (b) backward is linear in the output gradient: scaling d(logits) by 2 (exact in fp32 / bf16) doubles every
        parameter gradient exactly -- through every conv, BN, pool, fusion and the grouped weight-gradient reduction;
    (c) the training forward is reproducible run to run with the same dropout seed (no atomics in the forward path)."""
    from stcd_amd import synth

    a, b, _ = synth.make_batch(16, 256, 256, seed=77)
    A, B = t(a).to(DEV), t(b).to(DEV)
    torch.manual_seed(5)
    m = SiamUnet_diff(3, 2, dtype="bf16").to(DEV)
    m.eval()
    with torch.no_grad():
        full = m(A, B)
        h0, h1 = m(A[:8], B[:8]).clone(), m(A[8:], B[8:]).clone()
    assert torch.equal(full, torch.cat([h0, h1]))
    m.train()
    m.set_dropout_p(0.2)
    grads = []
    outs = []
    for scale in (1.0, 2.0):
        m._steps = 0                       # same dropout masks for both passes
        m.zero_grad(set_to_none=True)
        out = m(A, B)
        outs.append(out.detach().clone())
        g0 = torch.ones_like(out) * 1e-3
        out.backward(g0 * scale)
        grads.append(torch.cat([p.grad.flatten() for p in m.parameters()]).clone())
    assert torch.equal(outs[0], outs[1])
    assert torch.isfinite(grads[0]).all() and grads[0].abs().max() > 0
    # linear kernels + power-of-two scale: equal up to the summation order of the atomically folded slab parts and the
    # 2^-36 fixed-point step of the BatchNorm backward accumulators (rint(2x) != 2 rint(x))
    rel = ((grads[1] - 2.0 * grads[0]).abs().max() / grads[1].abs().max()).item()
    assert rel < 5e-5, rel


@pytest.mark.parametrize("label", [1, 2])
def test_snunet_bf16_tracks_reference_vectors(golden, label):
    """BASELINE config 3's arithmetic (SNUNet_ECAM, bf16 MFMA path incl. dense-concat slices, k2-s2 transposed convs, the
    1x1 head and ECAM) against the vectors captured from the reference: mean |dlogit| < 4e-2, max < 0.35, loss within 2e-2,
    gradient cosine vs the reference's gradient > 0.9 for every tensor of >= 256 elements outside the first encoder block
    (> 0.8 there: its gradient passes through every bf16 rounding of the 5-level net; on a 32x32 input the deepest
    BatchNorms normalise over 8 samples).  The production-shaped bound is the 128x128 test below."""
    from oracle import snunet_ref as S
    from stcd_amd.modules import SNUNet_ECAM
    from tests._util import ACHIEVED, gf_index, zero_grad_by_construction

    g = golden(f"g2_snunet_{label}.npz")
    seed = int(g["seed"])
    x1, x2 = t(g["x1"]).to(DEV), t(g["x2"]).to(DEV)
    m = SNUNet_ECAM(3, label, dtype="bf16")
    m.load_state_dict(S.synth_state(3, label, seed))
    m.to(DEV).train()
    logits = m(x1, x2)
    err = np.abs(logits.detach().cpu().numpy() - g["logits_train"])
    scale = max(1.0, float(np.abs(g["logits_train"]).max()))
    assert err.mean() < 4e-2 * scale and err.max() < 0.35 * scale, (err.mean(), err.max(), scale)
    loss = loss_fn(label, logits, t(g["target"]).to(DEV))
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    worst = {"rest": 1.0, "first": 1.0}
    for name, p in m.named_parameters():
        if zero_grad_by_construction(name) or p.numel() < 256:
            continue
        a = p.grad.flatten().cpu().double().numpy()[gf_index(name, p.numel())]
        _, cos = rel_l2_cos(a, g["gf/" + name])
        k = "first" if name.startswith("conv0_0.") else "rest"
        worst[k] = min(worst[k], cos)
        assert cos > (0.8 if k == "first" else 0.9), (name, cos)
    ACHIEVED[f"bf16 snunet({label}) vs reference G2 [worst cosine: all but conv0_0, conv0_0]"] = (worst["rest"], worst["first"])


@pytest.mark.parametrize("arch", ["conc", "snunet"])
def test_full_size_properties_other_configs_bf16(arch):
    """BASELINE configs 2 and 3 at their per-GPU size (SiamUnet_conc / SNUNet_ECAM, 16 pairs of 256x256, bf16) through the
    size-independent properties of test_full_size_properties_bf16: (a) eval mode treats pairs independently -- the
    16-pair batch equals its two 8-pair halves bit for bit (every conv tile walk, the dense-concat slices, ECAM's
    per-image pooling); (b) the backward is linear in d(logits): doubling it doubles every parameter gradient (up to the
    order of the atomically folded partial sums); (c) everything finite, gradients non-zero for every parameter tensor
    that can receive one."""
    from stcd_amd import synth
    from stcd_amd.modules import SNUNet_ECAM
    from tests._util import zero_grad_by_construction

    a, b, _ = synth.make_batch(16, 256, 256, seed=78)
    A, B = t(a).to(DEV), t(b).to(DEV)
    torch.manual_seed(6)
    m = (SNUNet_ECAM(3, 2, dtype="bf16") if arch == "snunet" else SiamUnet_conc(3, 2, dtype="bf16")).to(DEV)
    m.eval()
    with torch.no_grad():
        full = m(A, B).clone()
        h0, h1 = m(A[:8], B[:8]).clone(), m(A[8:], B[8:]).clone()
    assert torch.isfinite(full).all()
    assert torch.equal(full, torch.cat([h0, h1]))
    m.train()
    grads = []
    for scale in (1.0, 2.0):
        m._steps = 0                       # same dropout masks for both passes
        m.zero_grad(set_to_none=True)
        out = m(A, B)
        out.backward(torch.ones_like(out) * 1e-3 * scale)
        grads.append(torch.cat([p.grad.flatten() for p in m.parameters()]).clone())
        if scale == 1.0:
            for name, p in m.named_parameters():
                assert torch.isfinite(p.grad).all(), name
                if not zero_grad_by_construction(name):
                    assert p.grad.abs().max().item() > 0, name
    rel = ((grads[1] - 2.0 * grads[0]).abs().max() / grads[1].abs().max()).item()
    assert rel < 5e-5, rel


# bf16 bounds of the 128x128 step.  Forward: activations are STORED in bf16 between kernels (2^-9 relative per rounding, a
# random walk over the depth of the net): logits within 6 % of their mean magnitude, loss within 2e-2 (measured: 1.5-3.5 %,
# 3e-5..2e-3).  Gradients: the network is piecewise linear (ReLU gates, max-pool arg-max, |a1-a2| sign) and at this
# random-init state extremely flip-sensitive (tests/_util.py): rounding the FORWARD tensors of the CPU oracle to bf16 --
# fp32 arithmetic everywhere, gradients untouched -- already moves its gradients to a median cosine of 0.90 (diff) / 0.94
# (conc) against its own fp32 run, worst tensor 0.82 / 0.86 (tests/tools_bf16_emulation.py; rounding the BACKWARD tensors
# changes nothing: cosine 1.0000).  The engine measures median 0.88 / 0.91, worst 0.76 / 0.76; SNUNet (residual blocks, no
# |a-b| fusion) median 0.987, worst 0.952.  The bounds sit just under the measured values: what they catch is a wrong
# term, not rounding (the exact-arithmetic checks of the same kernels are the fp32 runs above and the per-op tests).
BF16_LOGIT_ERR, BF16_LOSS_ERR = 6e-2, 2e-2
BF16_GRAD = {"diff": (0.72, 0.80), "conc": (0.72, 0.80), "snunet": (0.94, 0.36), "fcef": (0.72, 0.80), "xconc": (0.70, 0.82)}      # (xconc: two more BatchNorm-ed layers on every skip; measured 0.7198 / 0.761)     # (min cosine, max relative l2) per tensor


def _g7_inputs(g, arch):
    seed = int(g["seed"])
    rng = np.random.default_rng(seed + 1)
    a = rng.standard_normal((2, 3, 128, 128)).astype(np.float32)
    b = (a + 0.5 * rng.standard_normal((2, 3, 128, 128))).astype(np.float32)
    tgt = t((np.random.default_rng(seed + 4).random((2, 128, 128)) < 0.2).astype(np.int64))
    if arch == "snunet":
        from oracle import snunet_ref as S
        from stcd_amd.modules import SNUNet_ECAM
        return seed, t(a), t(b), tgt, S.synth_state(3, 2, seed), None, SNUNet_ECAM
    return seed, t(a), t(b), tgt, R.synth_state(arch, 3, 2, seed), R.synth_masks(arch, 2, seed + 3), CLS[arch]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("arch", ["diff", "conc", "snunet", "fcef", "xconc"])
def test_train_step_128_tracks_reference_vectors(golden, arch, dtype):
    """G7: one train-mode step at 2 x 128 x 128 against the reference's logits, loss and EVERY parameter's gradient.
    fp32 engine: logits 1e-3, loss 1e-4, per-tensor relative l2 <= 2e-2 / cosine >= 0.9995.
    bf16 engine (the path the bench times -- MFMA kernels, fused statistics, grouped weight gradients): mean |dlogit|
    <= 6 % of mean |logit|, loss within 2e-2, per-tensor gradient cosine / relative l2 per BF16_GRAD (see the comment
    there: the forward rounding alone, emulated on the CPU oracle, explains the distance); the achieved values are
    printed in the test summary and recorded in DESIGN.md."""
    from tests._util import ACHIEVED, gf_index, zero_grad_by_construction
    g = golden(f"g7_{arch}_128.npz")
    seed, x1, x2, tgt, st, masks, cls = _g7_inputs(g, arch)
    m = cls(3, 2, dtype=dtype)
    m.load_state_dict(st)
    m.to(DEV).train()
    if masks is not None:
        m.set_dropout_masks(masks)
    logits = unwrap(m(x1.to(DEV), x2.to(DEV)))
    got = logits.detach().flatten().cpu().numpy()[g["logits_sample_idx"]]
    loss = torch.nn.functional.cross_entropy(logits, tgt.to(DEV))
    loss.backward()
    if dtype == "fp32":
        np.testing.assert_allclose(got, g["logits_sample"], rtol=1e-3, atol=1e-4)
        assert abs(loss.item() - float(g["loss"])) < 1e-4
        for name, p in m.named_parameters():
            check_grad(name, p.grad, g, tag=f"fp32 {arch} 128x128 step vs reference G7")
        return
    err = np.abs(got - g["logits_sample"]).mean() / float(g["logits_absmean"])
    dloss = abs(loss.item() - float(g["loss"]))
    worst = [0.0, 1.0, 0.0, 1.0]
    worst_name = ["", ""]
    table = {}
    for name, p in m.named_parameters():
        if zero_grad_by_construction(name) or p.numel() < 64:
            continue
        a = p.grad.flatten().cpu().double().numpy()[gf_index(name, p.numel())]
        rel, cos = rel_l2_cos(a, g["gf/" + name])
        table[name] = (round(rel, 4), round(cos, 5))
        first = name.startswith(("conv11.", "bn11.", "conv12.", "bn12.", "conv0_0."))
        k = 2 if first else 0
        if cos < worst[k + 1]:
            worst_name[k // 2] = name
        worst[k], worst[k + 1] = max(worst[k], rel), min(worst[k + 1], cos)
    import json, os
    if os.path.isdir("gpurun_out"):
        json.dump(table, open(f"gpurun_out/bf16_grad_parity_{arch}.json", "w"), indent=0)
    ACHIEVED[f"bf16 {arch} 128x128 step vs reference G7 [all but first block; worst {worst_name[0]}]"] = (worst[0], worst[1])
    ACHIEVED[f"bf16 {arch} 128x128 step vs reference G7 [first encoder block; worst {worst_name[1]}]"] = (worst[2], worst[3])
    ACHIEVED[f"bf16 {arch} 128x128 step vs reference G7 [mean |dlogit| / mean |logit|, |dloss|]"] = (float(err), dloss)
    assert err < BF16_LOGIT_ERR, err
    assert dloss < BF16_LOSS_ERR, (loss.item(), float(g["loss"]))
    cos_min, rel_max = BF16_GRAD[arch]
    assert worst[1] >= cos_min and worst[0] <= rel_max, (worst_name[0], worst[0], worst[1])
    assert worst[3] >= cos_min and worst[2] <= rel_max, (worst_name[1], worst[2], worst[3])


def test_frozen_weights_context_packs_once_and_is_exact():
    """Inference loops may vouch for constant weights (model.frozen_weights()): identical outputs to the plain path; a weight change
    INSIDE the context is by contract not seen until it ends; training-mode forwards always repack."""
    rng = np.random.default_rng(5)
    x1 = torch.from_numpy(rng.standard_normal((2, 3, 32, 32)).astype(np.float32)).to(DEV)
    x2 = torch.from_numpy(rng.standard_normal((2, 3, 32, 32)).astype(np.float32)).to(DEV)
    m = SiamUnet_diff(3, 2, dtype="bf16")
    m.load_state_dict(R.synth_state("diff", 3, 2, 3, perturb_running=True))
    m.to(DEV).eval()
    with torch.no_grad():
        plain = m(x1, x2).clone()
        with m.frozen_weights():
            a = m(x1, x2).clone()
            b = m(x2, x1).clone()
            assert torch.equal(a, plain)
            m.conv11.weight.mul_(2.0)                       # not noticed while frozen (documented)
            assert torch.equal(m(x1, x2), plain)
        changed = m(x1, x2).clone()                         # context over: repacked
        assert not torch.equal(changed, plain)
        with m.frozen_weights():                            # a new context packs the current weights
            assert torch.equal(m(x1, x2), changed)
            assert torch.equal(m(x2, x1), m(x2, x1))
    m.train()
    with m.frozen_weights():                                # training forwards ignore the vouching
        m.set_dropout_masks(R.synth_masks("diff", 2, 9))
        l1 = m(x1, x2).sum()
        with torch.no_grad():
            m.conv11.weight.mul_(0.5)
        m.set_dropout_masks(R.synth_masks("diff", 2, 9))
        l2 = m(x1, x2).sum()
        assert l1.item() != l2.item()


# ---------------------------------------------------------------------------------------------------------------------
# Layer-local parity inside the running FC-Siam networks (stcd_ws_tensor_* introspection): bf16 end-to-end gradients can only be
# bounded statistically (see BF16_GRAD above), so every Conv(Transpose)2d + BatchNorm + ReLU layer is checked IN PLACE against
# torch's own layer applied to the layer's stored input / output gradient: conv output (one bf16 rounding), weight and bias
# gradient (fp32 accumulation on both sides), BatchNorm + ReLU per date.  Dropout is switched off (p = 0) so the activation is a
# function of the stored conv output alone.
@pytest.mark.parametrize("arch,dtype,B,H,W,virt", [("diff", "bf16", 2, 64, 64, 0), ("conc", "bf16", 3, 48, 80, 0), ("sub", "bf16", 2, 32, 32, 0),
                                                   ("diff", "fp32", 2, 32, 48, 0), ("diff", "bf16", 2, 64, 64, 1), ("conc", "bf16", 3, 48, 80, 1)])
def test_fcsiam_every_layer_in_place(monkeypatch, arch, dtype, B, H, W, virt):
    monkeypatch.setenv("STCD_VIRT_ACT", str(virt))      # 1: the virtual-activation plan (opt-in)
    rng = np.random.default_rng(41)
    x1 = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
    x2 = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
    tgt = torch.from_numpy((rng.random((B, H, W)) < 0.2).astype(np.int64)).to(DEV)
    m = CLS[arch](3, 2, dtype=dtype)
    m.load_state_dict(R.synth_state(arch, 3, 2, 21))
    m.set_dropout_p(0.0)
    m.to(DEV).train()
    loss = torch.nn.functional.cross_entropy(unwrap(m(x1, x2)), tgt)
    loss.backward()
    torch.cuda.synchronize()
    ws = m._engine.ws_tensors()
    names = sorted({k.split(".")[0] for k in ws})
    assert len(names) == 19, names              # 10 encoder + 9 decoder conv+BN layers (conv11d has no BatchNorm: checked end to end)
    tol = 5e-6 if dtype == "fp32" else 3e-3      # bf16: the stored Y is rounded to bf16 (2^-9 per element)
    wq = (lambda w: w.detach().to(torch.bfloat16).float()) if dtype == "bf16" else (lambda w: w.detach())
    nchw = lambda t_: t_.permute(0, 3, 1, 2).float().contiguous()
    worst = {}

    def chk(kind, name, got, want, t_=tol):
        r = float((got.float() - want).norm() / want.norm().clamp_min(1e-30))
        if r >= worst.get(kind, (0.0, ""))[0]:
            worst[kind] = (r, name)
        assert r <= t_, (kind, name, r)

    def bn_relu(pname, y_all):
        """A = relu(BatchNorm_train(Y)) per BatchNorm group (encoder: one group per date), rounded as the engine stores / stages it"""
        pbn = getattr(m, "bn" + pname[4:])
        ng = 1 if pname.endswith("d") else 2
        n_ = y_all.shape[0] // ng
        outs = []
        for gi in range(ng):
            y = y_all[gi * n_:(gi + 1) * n_]
            mu, var = y.mean(dim=(0, 2, 3), keepdim=True), y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            outs.append(torch.relu((y - mu) * torch.rsqrt(var + 1e-5) * pbn.weight.detach().view(1, -1, 1, 1) + pbn.bias.detach().view(1, -1, 1, 1)))
        a = torch.cat(outs)
        return a.to(torch.bfloat16).float() if dtype == "bf16" else a

    n_virt = 0
    for name in names:
        conv, bn = getattr(m, name), getattr(m, "bn" + name[4:])
        if name + ".in.virt" in ws:
            # virtual activation (round 4): the layer reads its producer's RAW conv output and applies BN + ReLU (+ Dropout2d) while
            # staging -- in the forward launch and in the weight gradient.  Both are checked against torch's convolution of the
            # activation computed HERE from that raw tensor; the producer records no A of its own.
            raw = ws[name + ".in.virt"]
            prod = [k[:-2] for k in ws if k.endswith(".Y") and ws[k].data_ptr() == raw.data_ptr()]
            assert len(prod) == 1 and not any(k.startswith(prod[0] + ".A.g") for k in ws), (name, prod)
            X = bn_relu(prod[0], nchw(raw))
            n_virt += 1
        else:
            X = nchw(ws[name + ".in"])[:, :conv.in_channels]      # conv11: 3 of 8 padded channels
        Y, dY = nchw(ws[name + ".Y"]), nchw(ws[name + ".dY"])
        Wv = wq(conv.weight).requires_grad_(True)
        bv = conv.bias.detach().clone().requires_grad_(True)
        if isinstance(conv, torch.nn.ConvTranspose2d):
            Yt = torch.nn.functional.conv_transpose2d(X, Wv, bv, stride=1, padding=1)
        else:
            Yt = torch.nn.functional.conv2d(X, Wv, bv, padding=1)
        chk("conv output" + (" (virtual input)" if name + ".in.virt" in ws else ""), name, Y, Yt.detach())
        Yt.backward(dY)
        # fp32 accumulation of the same bf16 products on both sides; a virtual input is re-derived HERE from the raw tensor (torch's
        # batch statistics against the engine's fixed-point sums): a handful of activations round to the neighbouring bf16 value
        chk("weight gradient" + (" (virtual input)" if name + ".in.virt" in ws else ""), name, conv.weight.grad, Wv.grad, 2e-4 if name + ".in.virt" in ws else 5e-6)
        # a bias in front of a training-mode BatchNorm has a mathematically zero gradient (the per-date sums of dY vanish): both
        # sides are rounding noise, so bound them against the natural scale sum|dY| instead of against each other
        scale = float(dY.abs().sum(dim=(0, 2, 3)).max())
        for side, gb in (("engine", conv.bias.grad), ("torch", bv.grad)):
            r = float(gb.abs().max()) / scale
            worst["bias gradient / sum|dY|"] = max(worst.get("bias gradient / sum|dY|", (0.0, "")), (r, name + ":" + side))
            assert r <= (1e-5 if dtype == "fp32" else 4e-3), (name, side, r)
        groups = [k for k in ws if k.startswith(name + ".A.g")]
        if not groups:
            continue                               # a virtual layer: its activation exists only inside its consumer's staging (checked there)
        npg = Y.shape[0] // len(groups)
        for gi in range(len(groups)):
            y = Y[gi * npg:(gi + 1) * npg]
            mu, var = y.mean(dim=(0, 2, 3), keepdim=True), y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            a = torch.relu((y - mu) * torch.rsqrt(var + 1e-5) * bn.weight.detach().view(1, -1, 1, 1) + bn.bias.detach().view(1, -1, 1, 1))
            chk("bn+relu", name, nchw(ws[f"{name}.A.g{gi}"]), a, 5e-6 if dtype == "fp32" else 4e-3)
    if dtype == "bf16" and virt:
        assert n_virt >= 9, n_virt          # 6 encoder + >= 3 decoder layers read a virtual activation in the bf16 plan
    else:
        assert n_virt == 0                  # the default plan and the fp32 parity path materialise everything
    print(f"SiamUnet_{arch} {dtype} B={B} {H}x{W} layer-local worst relative l2: " + ", ".join(f"{k} {v[0]:.1e} ({v[1]})" for k, v in worst.items()))


@pytest.mark.parametrize("arch", ["diff", "conc", "fcef", "snunet", "changeformer"])
def test_side_stream_weight_gradients_are_deterministic_and_match_the_serial_plan(arch):
    """FC-Siam backward in one call: the decoder's grouped weight gradients run on the engine's low-priority side stream beside the
    encoder's chain (stcd_set_wgrad_side, on by default) with a quarter of the planner's block budget.  (1) Race screen: every
    reduction of the engine is fixed-order since round 4 (slab sums finish inside one block in index order, bias / BatchNorm sums in
    integer accumulators; rounds 1-3 finished many-slab sums with float atomics and this screen was a 1e-5 bound), so 12 backward
    passes of one state / batch at the headline size must be BIT-IDENTICAL to the first, with allocator / cache noise in between --
    a weight gradient read before its dY is final, or a slab summed before it is written, shows as any difference at all.
    (2) The same gradients from the serial plan (stcd_set_wgrad_side(0): more K-split slabs, i.e.
    another fp32 summation order over bf16 products) agree to fp32 summation noise, tensor by tensor."""
    torch.manual_seed(3)
    B, H, W = 16, 256, 256
    if arch == "changeformer":  # V6 at 4 x 512^2 (BASELINE.json configs[4] per GPU): the head's k_wgrad_dma group (192 blocks beside the chain
        B, H, W = 4, 512, 512   # instead of 256 alone) goes out after the second up-sampling layer's backward
        from stcd_amd.changeformer import ChangeFormerV6
    if arch == "snunet":      # one backward stage: the groups whose members sit deep in the backward order run beside the rest of the chain
        from oracle import snunet_ref as S
        from stcd_amd.modules import SNUNet_ECAM
    cls = {"diff": SiamUnet_diff, "conc": SiamUnet_conc, "fcef": Unet}.get(arch) or (SNUNet_ECAM if arch == "snunet" else ChangeFormerV6)
    x1 = torch.randn(B, 3, H, W, device=DEV); x2 = torch.randn(B, 3, H, W, device=DEV)
    tgt = (torch.rand(B, H, W, device=DEV) < 0.2).long()
    st = None if arch == "changeformer" else (R.synth_state(arch, 3, 2, 5) if arch != "snunet" else S.synth_state(3, 2, 5))
    if st is None:            # its own (seeded) initialisation; every engine below loads the same state
        torch.manual_seed(11)
        st = {k: v.clone() for k, v in cls(3, 2, dtype="bf16").state_dict().items()}

    def grads(side, reps):
        m = cls(3, 2, dtype="bf16")
        m.load_state_dict(st)
        m.to(DEV).train()
        m._engine.set_wgrad_side(side)
        outs = []
        junk = torch.empty(96 << 20, dtype=torch.uint8, device=DEV)
        for it in range(reps):
            m.zero_grad(set_to_none=False)
            m._steps = 0                                     # the same dropout masks every pass (ChangeFormer: the same hash seed)
            out = m(x1, x2)
            out = out[-1] if isinstance(out, (list, tuple)) else out
            torch.nn.functional.cross_entropy(out, tgt).backward()
            outs.append(m._flat_grads.clone())
            if it % 2:
                junk.fill_(it)
        torch.cuda.synchronize()
        return outs

    on = grads(True, 12)
    off = grads(False, 1)[0]
    infos = cls(3, 2, dtype="bf16")._engine.params
    worst = [0.0, 0.0]
    for info in infos:
        sl = slice(info.offset, info.offset + info.numel)
        a = on[0][sl].double()
        for k, g in enumerate(on[1:], 1):
            rel = float((g[sl].double() - a).norm() / (a.norm() + 1e-30))
            worst[0] = max(worst[0], rel)
            assert torch.equal(g[sl], on[0][sl]), f"{arch} {info.name}: backward pass {k} differs from pass 0 (rel-l2 {rel:.2e}) with the side stream on: the engine must be bit-reproducible"
        b = off[sl].double()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        worst[1] = max(worst[1], rel)
        assert rel <= (2e-5 if arch != "changeformer" else 1e-4), f"{arch} {info.name}: side-stream plan vs serial plan rel-l2 {rel:.2e}"
    print(f"{arch}: worst rel-l2 between passes {worst[0]:.2e}, side-stream plan vs serial plan {worst[1]:.2e}")


@pytest.mark.parametrize("arch,B,H,W", [("diff", 4, 64, 64), ("conc", 2, 48, 80), ("sub", 3, 32, 32), ("fcef", 2, 64, 64), ("diff", 2, 100, 100),
                                        ("diff", 16, 256, 256)])
def test_virtual_activations_equal_the_materialised_plan_bit_for_bit(monkeypatch, arch, B, H, W):
    """Round 4: a non-skip conv -> BN -> ReLU -> Dropout2d layer whose only reader is the next conv of its stage never writes its
    activation -- the reader's forward launch (k_conv_small / k_conv_res, XF variants) and its weight gradient (k_wgrad_group,
    job.xf_*) apply scale / shift / ReLU / mask while staging the raw conv output (common.h XfSrc / xf_act8, the arithmetic of
    k_bn_act).  The plan with every activation materialised (STCD_VIRT_ACT=0, rounds 1-3) must give the SAME bits: logits, loss,
    every parameter gradient, BatchNorm running statistics (the consumer's block 0 publishes them now), in training mode with
    Dropout2d active (odd sizes and the headline size included), and the eval-mode logits (running statistics path).
    MEASURED: the virtual plan is 2-3 % SLOWER on the headline step (every consumer launch pays more
    in its loader and prologue than the removed k_bn_act cost), so it is opt-in (STCD_VIRT_ACT=1) and the default plan materialises."""
    torch.manual_seed(5)
    x1 = torch.randn(B, 3, H, W, device=DEV); x2 = torch.randn(B, 3, H, W, device=DEV)
    tgt = (torch.rand(B, H, W, device=DEV) < 0.2).long()
    st = R.synth_state(arch, 3, 2, 9)
    res = []
    # (the virtual plan keeps every k_bn_reduce launch; the materialised default forms three layers' backward sums inside their data
    #  gradients -- another summation order -- so both plans run with that fusion off for the bit-for-bit comparison)
    monkeypatch.setenv("STCD_NO_BWDSUM_FUSE", "1")
    for virt in ("1", "0"):
        monkeypatch.setenv("STCD_VIRT_ACT", virt)
        m = CLS[arch](3, 2, dtype="bf16")
        m.load_state_dict(st)
        m.to(DEV).train()
        m._seed, m._steps = 123, 0                      # the same Dropout2d masks in both plans
        out = unwrap(m(x1, x2))
        loss = torch.nn.functional.cross_entropy(out, tgt)
        loss.backward()
        torch.cuda.synchronize()
        nvirt = sum(1 for k in m._engine.ws_tensors() if k.endswith(".in.virt"))
        m.eval()
        with torch.no_grad():
            ev = unwrap(m(x1, x2)).clone()
        res.append((out.detach().clone(), loss.item(), m._flat_grads.clone(), m._flat_bn.clone(), ev, nvirt))
    a, b = res
    assert a[5] >= 9 and b[5] == 0, (a[5], b[5])
    assert torch.equal(a[0], b[0]), float((a[0] - b[0]).abs().max())
    assert abs(a[1] - b[1]) <= 1e-6 * abs(b[1])      # (torch's own cross_entropy reduction: float atomics, not ours)
    # gradients: the data path is bit-identical; the weight gradients of the layers that read a virtual activation run in their own
    # grouped grid (k_wgrad_group<.., XF>), whose K-split may differ from the shared grid's at large sizes: another fp32 summation
    # order of the SAME bf16 products (bit-equal at the small sizes, <= 1e-6 relative at the headline size)
    rel = float((a[2].double() - b[2].double()).norm() / b[2].double().norm())
    assert rel <= 1e-6, rel
    if B * H * W <= 4 * 64 * 64 * 2:
        assert torch.equal(a[2], b[2]), float((a[2] - b[2]).abs().max())
    assert torch.equal(a[3], b[3]), float((a[3] - b[3]).abs().max())
    assert torch.equal(a[4], b[4]), float((a[4] - b[4]).abs().max())


@pytest.mark.parametrize("arch,B,H,W,dtype", [("diff", 4, 64, 64, "bf16"), ("sub", 3, 32, 32, "bf16"), ("diff", 2, 100, 100, "bf16"),
                                              ("diff", 3, 33, 47, "bf16"), ("diff", 2, 64, 64, "fp32"), ("sub", 2, 50, 34, "fp32"),
                                              ("diff", 16, 256, 256, "bf16")])
def test_recomputed_skip_activations_equal_the_stored_plan(monkeypatch, arch, B, H, W, dtype):
    """Round 4: the skip layers of diff / sub (last conv of an encoder level) do not store their activations -- k_bn_act_pair writes
    only the pooled map and the fused skip |a1 - a2| / a2 - a1, and the backward (k_skip_bwd_pair) recomputes a = round(max(fma(y,
    scale * mk, shift * mk), 0)) from the conv output with the forward's arithmetic, both dates of a pair in one thread.  The plan
    that stores them (STCD_NO_SKIP_RECOMPUTE=1: k_bn_act_pair<STORE_A> + the per-date k_skip_bwd of rounds 2-3) must give the SAME
    bits: logits, every gradient, the running statistics -- with Dropout2d active, odd sizes (unpooled border cells) included."""
    torch.manual_seed(6)
    x1 = torch.randn(B, 3, H, W, device=DEV); x2 = torch.randn(B, 3, H, W, device=DEV)
    tgt = (torch.rand(B, H, W, device=DEV) < 0.2).long()
    st = R.synth_state(arch, 3, 2, 11)
    res = []
    for stored in ("0", "1"):
        monkeypatch.setenv("STCD_NO_SKIP_RECOMPUTE", stored)
        m = CLS[arch](3, 2, dtype=dtype)
        m.load_state_dict(st)
        m.to(DEV).train()
        m._seed, m._steps = 77, 0
        out = unwrap(m(x1, x2))
        torch.nn.functional.cross_entropy(out, tgt).backward()
        torch.cuda.synchronize()
        nA = sum(1 for k in m._engine.ws_tensors() if ".A.g" in k)
        res.append((out.detach().clone(), m._flat_grads.clone(), m._flat_bn.clone(), nA))
    a, b = res
    assert b[3] - a[3] == 8, (a[3], b[3])            # four skip layers x two dates are not materialised
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[2], b[2])
    # gradients: dA is the same bits (tests/test_ew_ops_gpu.py checks that per op, and the sums to 1e-6); the BatchNorm partial sums
    # are the same per-thread floats, summed exactly per block and rounded to float once -- but the two kernels cut the map into
    # different blocks, so the level's sums differ by float roundings (~1e-7).  In fp32 mode that is where it stays; in bf16 mode every
    # re-rounding of dY that such a difference flips downstream is a 4e-3 step on that element, so the encoder's gradients of two
    # equally valid plans end up as far apart as two bf16 roundings of the same numbers (measured 1e-9 ... 5e-4 over all parameters,
    # 3e-6 at conv43 growing to 3.6e-3 at conv11 at the headline size)
    rel = float((a[1].double() - b[1].double()).norm() / b[1].double().norm())
    print(f"recomputed vs stored skip activations {arch} {B}x{H}x{W} {dtype}: gradient rel-l2 {rel:.2e}")
    assert rel <= (2e-3 if dtype == "bf16" else 1e-5), rel


@pytest.mark.parametrize("arch,B,H,W", [("diff", 4, 64, 64), ("conc", 2, 64, 96), ("fcef", 4, 128, 128), ("diff", 2, 100, 100)])
def test_fused_backward_sums_equal_the_separate_launches(monkeypatch, arch, B, H, W):
    """Round 4: a data gradient that writes dA of a conv -> BN -> ReLU -> Dropout2d layer can also form that layer's BatchNorm-backward
    sums (sum(dz), sum(dz * xhat) of the rounded dA it stores; k_conv_small<.., BWD> by default for the two level-1 layers,
    k_conv_res<.., BWD> with STCD_BWDSUM_RES=1 for the deeper ones) -- the layer's k_bn_reduce launch disappears.  Against the plan
    with every k_bn_reduce launch (STCD_NO_BWDSUM_FUSE=1): the same logits and running statistics bit for bit; the gradients as
    close as two summation orders of the same numbers leave bf16 training (see test_recomputed_skip_activations_equal_the_stored_plan)."""
    torch.manual_seed(9)
    x1 = torch.randn(B, 3, H, W, device=DEV); x2 = torch.randn(B, 3, H, W, device=DEV)
    tgt = (torch.rand(B, H, W, device=DEV) < 0.2).long()
    st = R.synth_state(arch, 3, 2, 13)
    res = []
    for env in ({"STCD_NO_BWDSUM_FUSE": "1"}, {}, {"STCD_BWDSUM_RES": "1"}):
        monkeypatch.delenv("STCD_NO_BWDSUM_FUSE", raising=False)
        monkeypatch.delenv("STCD_BWDSUM_RES", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = CLS[arch](3, 2, dtype="bf16")
        m.load_state_dict(st)
        m.to(DEV).train()
        m._seed, m._steps = 31, 0
        out = unwrap(m(x1, x2))
        torch.nn.functional.cross_entropy(out, tgt).backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), m._flat_grads.clone(), m._flat_bn.clone()))
    ref = res[0]
    for name, r in zip(("k_conv_small layers", "k_conv_small + k_conv_res layers"), res[1:]):
        assert torch.equal(r[0], ref[0]) and torch.equal(r[2], ref[2]), name
        rel = float((r[1].double() - ref[1].double()).norm() / ref[1].double().norm())
        print(f"fused backward sums ({name}) vs separate launches, {arch} {B}x{H}x{W}: gradient rel-l2 {rel:.2e}")
        # (measured 1e-8 ... 4e-3: the sums differ by ~1e-8; what grows from there is the bf16 re-rounding of dY, x3-5 per layer where
        #  the deepest BatchNorms see only a few hundred values per channel -- FC-EF at 2 x 64 x 64: 1e-8 at bn12d, 6e-3 at conv42d)
        assert rel <= 1e-2, (name, rel)
