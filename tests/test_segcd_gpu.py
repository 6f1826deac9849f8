"""SegCD (ResNet-50 UNet, the model the reference's scripts train) on the HIP engine, through the nn.Module boundary ->
C ABI, against the vectors captured from the reference's own ResNet / UnetDecoder / SegmentationHead (G10) and against the
CPU oracle (oracle/segcd_ref.py) on fresh inputs."""
import numpy as np
import pytest
import torch

from oracle import segcd_ref as G
from stcd_amd.losses import bce_dice_with_logits
from stcd_amd.segcd import FFCTLCD, SegCD, UnetSeg
from tests._util import check_grad, rel_l2_cos, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# G10's oracle-vs-reference bound (tests/test_oracle_golden.py): ~110 piecewise-linear layers and BatchNorms over as few as 8
# samples at the fixture's size -- two fp32 evaluation orders of the same arithmetic already differ by 2e-2 at the stem
SEG_REL, SEG_COS = 5e-2, 0.998


def _loss(m1, m2, ch, seg, tgt):
    return bce_dice_with_logits(m1, seg) + bce_dice_with_logits(ch, tgt) + 0.5 * m2.mean()


FIXTURES = [("g10_segcd.npz", 1, "resnet50"), ("g11_segcd_2cls.npz", 2, "resnet50"), ("g12_segcd_r18.npz", 1, "resnet18"),
            ("g13_segcd_r34.npz", 2, "resnet34"), ("g14_segcd_r101.npz", 1, "resnet101")]


@pytest.mark.parametrize("fixture,classes,encoder", FIXTURES)
def test_segcd_fp32_matches_reference_vectors(golden, fixture, classes, encoder):
    # G10: 1 class, 2 x 64 x 64; G11: 2 classes, 3 x 96 x 64; G12 / G13: the BasicBlock encoders; G14: resnet101, 2 x 128 x 128
    # -- all from the reference's own classes
    g = golden(fixture)
    seed = int(g["seed"])
    deep = encoder in ("resnet101", "resnet152")          # tests/test_oracle_golden.py: two fp32 orders differ by 6e-2 at this depth
    x1, x2 = t(g["x1"]).to(DEV), t(g["x2"]).to(DEV)
    m = SegCD(encoder_name=encoder, classes=classes, dtype="fp32")
    m.load_state_dict(G.synth_state(3, classes, seed, perturb_running=True, encoder=encoder))
    m.to(DEV).eval()
    with torch.no_grad():
        o = m(x1, x2)
    for k, v in zip(("m1", "m2", "change"), o):       # relative to the map's scale (resnet101's eval maps reach 4e3 with the synthetic statistics)
        np.testing.assert_allclose(v.cpu().numpy(), g[f"eval/{k}"], rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(g[f"eval/{k}"]).max())), err_msg=k)

    m = SegCD(encoder_name=encoder, classes=classes, dtype="fp32")
    m.load_state_dict(G.synth_state(3, classes, seed, encoder=encoder))
    m.to(DEV).train()
    m1, m2, ch = m(x1, x2)
    for k, v in zip(("m1", "m2", "change"), (m1, m2, ch)):
        np.testing.assert_allclose(v.detach().cpu().numpy(), g[f"train/{k}"], rtol=2e-3, atol=2e-3, err_msg=k)
    loss = _loss(m1, m2, ch, t(g["seg_target"]).to(DEV), t(g["target"]).to(DEV))
    assert abs(loss.item() - float(g["loss"])) < 2e-4
    loss.backward()
    for name, p in m.named_parameters():
        if float(np.abs(g["gs/" + name][1])) < 1e-12:
            continue
        check_grad(name, p.grad, g, rel_max=1.2e-1 if deep else SEG_REL, cos_min=0.992 if deep else SEG_COS, tag=f"fp32 SegCD vs reference {fixture[:3].upper()}")      # deep: worst seen 8.2e-2 / 0.9967 (oracle vs reference: 6.0e-2)
    sd = m.state_dict()
    for k in [k for k in g if k.startswith("rs/")]:     # layer4's inputs already differ by ~1e-3 relative between two fp32 evaluation orders
        np.testing.assert_allclose(sd[k[3:]].cpu().numpy(), g[k], rtol=1e-4, atol=5e-5, err_msg=k)


def test_segcd_fp32_at_the_north_star_bar_on_a_well_conditioned_fixture(golden):
    """G18 (round-2 review, weak #3): the reference's SegCD(resnet50) at 4 x 128 x 128 -- 64 samples per channel at the deepest
    BatchNorm instead of G10's 8.  With the fixture's conditioning out of the way the fp32 engine meets the north_star's 1e-3 on
    every map (train mode included) and per-tensor gradient bars of 3.5e-2 / 0.9994 (G10: 5e-2 / 0.998)."""
    g = golden("g18_segcd_128.npz")
    seed = int(g["seed"])
    x1, x2 = t(g["x1"]).to(DEV), t(g["x2"]).to(DEV)
    m = SegCD(encoder_name="resnet50", classes=1, dtype="fp32")
    m.load_state_dict(G.synth_state(3, 1, seed, perturb_running=True))
    m.to(DEV).eval()
    with torch.no_grad():
        o = m(x1, x2)
    for k, v in zip(("m1", "m2", "change"), o):
        np.testing.assert_allclose(v.cpu().numpy(), g[f"eval/{k}"], rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(g[f"eval/{k}"]).max())), err_msg=k)
    m = SegCD(encoder_name="resnet50", classes=1, dtype="fp32")
    m.load_state_dict(G.synth_state(3, 1, seed))
    m.to(DEV).train()
    m1, m2, ch = m(x1, x2)
    worst = 0.0
    for k, v in zip(("m1", "m2", "change"), (m1, m2, ch)):
        worst = max(worst, float(np.abs(v.detach().cpu().numpy() - g[f"train/{k}"]).max()))
        np.testing.assert_allclose(v.detach().cpu().numpy(), g[f"train/{k}"], rtol=1e-3, atol=1e-3, err_msg=k)
    loss = _loss(m1, m2, ch, t(g["seg_target"]).to(DEV), t(g["target"]).to(DEV))
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    for name, p in m.named_parameters():
        if float(np.abs(g["gs/" + name][1])) < 1e-12:
            continue
        # 3.5e-2 / 0.9994: the pinned CPU oracle itself sits at 2.68e-2 / 0.999642 against this fixture (stem weight: ~110 piecewise-linear
        # layers between it and the loss; tests/test_oracle_golden.py) -- the engine measures 3.26e-2 / 0.999474 on its worst tensor (the
        # stem's bn1.bias; 2.65e-2 / 0.999648 with an earlier summation order of the same kernels), every other tensor < 2.8e-2
        check_grad(name, p.grad, g, rel_max=3.5e-2, cos_min=0.9994, tag="fp32 SegCD vs reference G18 (4 x 128 x 128)")
    print(f"G18: worst train-mode map error {worst:.2e}")


@pytest.mark.parametrize("B,H,W,cin", [(1, 128, 160, 3), (3, 96, 64, 3), (3, 96, 96, 5)])
def test_segcd_fp32_matches_oracle_on_other_shapes(B, H, W, cin):
    """Odd batch, non-square sizes, 2 classes against the CPU oracle run in fp64 (gradients of the whole step).  Sizes keep >= 18
    samples per BatchNorm at the deepest stage (H/32 x W/32 x B): below that the normalisation amplifies fp32 rounding beyond any
    meaningful bound (3 samples at 1x32x96 move the outputs by 9e-2)."""
    seed = 77 + B
    rng = np.random.default_rng(seed)
    x1 = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    x2 = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    w1, w2, w3 = (torch.from_numpy(rng.standard_normal((B, 2, H, W)).astype(np.float32)) for _ in range(3))
    st = G.synth_state(cin, 2, seed)
    m = SegCD(encoder_name="resnet50", in_channels=cin, classes=2, dtype="fp32")
    m.load_state_dict(st)
    m.to(DEV).train()
    o = m(x1.to(DEV), x2.to(DEV))
    loss = sum((a * b.to(DEV)).sum() for a, b in zip(o, (w1, w2, w3))) / (B * H * W)
    loss.backward()
    st64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in st.items()}
    params = [k for k, v in st64.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        st64[k].requires_grad_(True)
    ro = G.forward(st64, x1.double(), x2.double(), training=True)
    rloss = sum((a * b.double()).sum() for a, b in zip(ro, (w1, w2, w3))) / (B * H * W)
    rloss.backward()
    for a, b, k in zip(o, ro, ("m1", "m2", "change")):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().float().numpy(), rtol=2e-3, atol=2e-3, err_msg=k)
    worst = (0.0, 1.0)
    for name, p in m.named_parameters():
        ref = st64[name].grad
        if ref is None or float(ref.abs().max()) < 1e-12:
            continue
        r, c = rel_l2_cos(p.grad.cpu().double().numpy(), ref.numpy())
        if r > worst[0]:
            worst_name = name
        worst = (max(worst[0], r), min(worst[1], c))
    print(f"worst tensor: {worst_name}")
    assert worst[0] <= SEG_REL and worst[1] >= SEG_COS, (worst_name, worst)
    print(f"SegCD fp32 vs fp64 oracle B={B} {H}x{W} cin={cin}: worst rel-l2 {worst[0]:.2e}, worst cos {worst[1]:.6f}")
    sd = m.state_dict()
    for k in ("encoder.bn1", "encoder.layer2.0.downsample.1", "decoder.blocks.3.conv2.1"):
        np.testing.assert_allclose(sd[k + ".running_mean"].cpu().numpy(), st64[k + ".running_mean"].float().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), st64[k + ".running_var"].float().numpy(), rtol=1e-4, atol=1e-5)
        assert int(sd[k + ".num_batches_tracked"]) == 2


def test_segcd_rejects_unsupported_configs_and_sizes():
    with pytest.raises(NotImplementedError):
        SegCD(encoder_name="resnext50_32x4d")
    with pytest.raises(NotImplementedError):
        SegCD(encoder_name="resnet34", decoder_attention_type="scse")
    m = SegCD(encoder_name="resnet50", dtype="fp32").to(DEV)
    with pytest.raises(Exception, match="divisible by 32"):
        m(torch.zeros(1, 3, 48, 48, device=DEV), torch.zeros(1, 3, 48, 48, device=DEV))


# ---------------------------------------------------------------------------------------------------------------------
# Layer-local parity INSIDE the network: rounding differences grow through ~110 BatchNorm-ed layers (with the synthetic
# weights the step's gradient changes by 3e-2 between two fp32 evaluation orders, and bf16 storage decorrelates it), so an
# end-to-end comparison cannot bound a bf16 kernel.  Instead every layer is checked in place against torch's convolution
# applied to the layer's OWN stored input / output gradient (stcd_ws_tensor_* introspection): output, weight gradient,
# input gradient, BatchNorm + ReLU (+ residual), and the concat / up-sampling plumbing -- at per-op tolerance.
def _nchw(t):
    return t.permute(0, 3, 1, 2).float().contiguous()


def _holders(m):
    """conv name -> (conv holder, bn holder) for every Conv+BN layer of the plan."""
    out = {"encoder.conv1": (m.encoder.conv1, m.encoder.bn1)}
    for li in range(1, 5):
        for b, blk in enumerate(getattr(m.encoder, f"layer{li}")):
            pre = f"encoder.layer{li}.{b}"
            for k in (1, 2, 3) if hasattr(blk, "conv3") else (1, 2):
                out[f"{pre}.conv{k}"] = (getattr(blk, f"conv{k}"), getattr(blk, f"bn{k}"))
            if blk.downsample is not None:
                out[f"{pre}.downsample.0"] = (blk.downsample[0], blk.downsample[1])
    for i, blk in enumerate(m.decoder.blocks):
        out[f"decoder.blocks.{i}.conv1.0"] = (blk.conv1[0], blk.conv1[1])
        out[f"decoder.blocks.{i}.conv2.0"] = (blk.conv2[0], blk.conv2[1])
    return out


@pytest.mark.parametrize("dtype,B,H,W,cin,encoder", [("fp32", 2, 64, 64, 3, "resnet50"), ("bf16", 2, 64, 96, 3, "resnet50"), ("bf16", 3, 128, 128, 3, "resnet50"),
                                                      ("bf16", 2, 96, 64, 6, "resnet50"), ("bf16", 1, 32, 64, 3, "resnet50"), ("fp32", 5, 32, 32, 1, "resnet50"),
                                                      ("fp32", 2, 64, 64, 3, "resnet18"), ("bf16", 3, 96, 128, 3, "resnet18"), ("bf16", 2, 64, 64, 4, "resnet34"),
                                                      ("bf16", 2, 64, 96, 3, "resnet101"), ("bf16", 1, 64, 64, 3, "resnet152")])
def test_segcd_every_layer_in_place(dtype, B, H, W, cin, encoder):
    tol = 2e-4 if dtype == "fp32" else 6e-3            # bf16: one rounding of the output (2^-9 relative, ~1.2e-3 rms) + bf16 weights
    rng = np.random.default_rng(31)
    x1 = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32)).to(DEV)
    x2 = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32)).to(DEV)
    tgt = torch.from_numpy((rng.random((B, 1, H, W)) < 0.2).astype(np.float32)).to(DEV)
    m = SegCD(encoder_name=encoder, in_channels=cin, dtype=dtype)
    m.load_state_dict(G.synth_state(cin, 1, 11, encoder=encoder))
    m._engine.set_debug(1)
    m.to(DEV).train()
    o = m(x1, x2)
    loss = bce_dice_with_logits(o[2], tgt) + bce_dice_with_logits(o[0], tgt) + 0.5 * o[1].mean()
    loss.backward()
    torch.cuda.synchronize()
    ws = m._engine.ws_tensors()
    hold = _holders(m)
    basic = G.ENCODERS[encoder][0] == 1
    nblk = sum(G.ENCODERS[encoder][1])
    assert len(hold) == 1 + (2 if basic else 3) * nblk + (3 if basic else 4) + 10 and all(f"{k}.Y" in ws for k in hold)      # resnet50: 63
    last = {li: f"encoder.layer{li}.{G.ENCODERS[encoder][1][li - 1] - 1}.conv{2 if basic else 3}" for li in (1, 2, 3, 4)}
    worst = {}

    def chk(kind, name, got, want, t=tol):
        r = float((got.float() - want).norm() / want.norm().clamp_min(1e-30))
        worst[kind] = max(worst.get(kind, (0.0, ""))[0], r), (name if r >= worst.get(kind, (0.0, ""))[0] else worst[kind][1])
        assert r <= t, (kind, name, r)

    wq = (lambda w: w.detach().to(torch.bfloat16).float()) if dtype == "bf16" else (lambda w: w.detach())
    for name, (conv, bn) in hold.items():
        X, Y, A, dY = _nchw(ws[name + ".in"]), _nchw(ws[name + ".Y"]), _nchw(ws[name + ".A"]), _nchw(ws[name + ".dY"])
        Wt = wq(conv.weight)
        if name == "encoder.conv1":
            X = X[:, :cin]
        chk("conv output", name, Y, torch.nn.functional.conv2d(X, Wt, None, conv.stride, conv.padding))
        # weight gradient from the stored input and output gradient (fp32 accumulation of bf16 products on both sides)
        want_dw = torch.nn.grad.conv2d_weight(X, conv.weight.shape, dY, conv.stride, conv.padding)
        chk("weight gradient", name, conv.weight.grad, want_dw, 2e-4 if dtype == "fp32" else 2e-3)
        if name + ".dIn" in ws:
            want_dx = torch.nn.grad.conv2d_input(X.shape, Wt, dY, conv.stride, conv.padding)
            if name == "encoder.layer1.0.conv1" and not basic:      # the max-pool's gradient buffer: the down-sample branch is accumulated into it
                dc = hold["encoder.layer1.0.downsample.0"][0]
                want_dx = want_dx + torch.nn.grad.conv2d_input(X.shape, wq(dc.weight), _nchw(ws["encoder.layer1.0.downsample.0.dY"]), dc.stride, dc.padding)
            # BasicBlock layer1.0 has no down-sample: the max-pool's gradient buffer also holds the gated residual gradient of
            # conv2's BatchNorm, which is dY(conv2) / scale ... not a stored tensor -- its input gradient is covered end to end
            if not (name == "encoder.layer1.0.conv1" and basic):
                chk("input gradient", name, _nchw(ws[name + ".dIn"]), want_dx)
        # BatchNorm (per-date batch statistics) + residual + ReLU from the stored conv output
        res = _nchw(ws[name + ".res"]) if name + ".res" in ws else None
        outs = []
        for d in range(2):
            y = Y[d * B:(d + 1) * B]
            mu, var = y.mean(dim=(0, 2, 3), keepdim=True), y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            z = (y - mu) * torch.rsqrt(var + 1e-5) * bn.weight.detach().view(1, -1, 1, 1) + bn.bias.detach().view(1, -1, 1, 1)
            if res is not None:
                z = z + res[d * B:(d + 1) * B]
            outs.append(z if ".downsample." in name else torch.relu(z))
        chk("bn+relu", name, A, torch.cat(outs), 1e-4 if dtype == "fp32" else 1.2e-2)
    # plumbing: max-pool, residual wiring, up-sampling + skip concat (exact copies of stored tensors)
    f1 = _nchw(ws["encoder.conv1.A"])
    assert torch.equal(_nchw(ws["encoder.layer1.0.conv1.in"]), torch.nn.functional.max_pool2d(f1, 3, 2, 1))
    cl = "conv2" if basic else "conv3"
    assert torch.equal(_nchw(ws[f"encoder.layer1.1.{cl}.res"]), _nchw(ws[f"encoder.layer1.0.{cl}.A"]))
    assert torch.equal(_nchw(ws[f"encoder.layer2.0.{cl}.res"]), _nchw(ws["encoder.layer2.0.downsample.0.A"]))
    if basic:       # layer1.0 has no down-sample: its residual is the max-pool's output
        assert torch.equal(_nchw(ws["encoder.layer1.0.conv2.res"]), _nchw(ws["encoder.layer1.0.conv1.in"]))
    skips = [last[3], last[2], last[1], "encoder.conv1"]
    x = _nchw(ws[last[4] + ".A"])
    for i in range(5):
        cat = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
        if i < 4:
            cat = torch.cat([cat, _nchw(ws[skips[i] + ".A"])], dim=1)
        assert torch.equal(_nchw(ws[f"decoder.blocks.{i}.conv1.0.in"]), cat), i
        x = _nchw(ws[f"decoder.blocks.{i}.conv2.0.A"])
    # head: logits from the last decoder activation, in torch
    d1, d2 = x[:B], x[B:]
    hw, hb = wq(m.segmentation_head[0].weight), m.segmentation_head[0].bias.detach()
    head = lambda t_: torch.nn.functional.conv2d(t_, hw, hb, padding=1)
    m1, m2 = head(d1), head(d2)
    want = (m1, m2, torch.min(head((d1 - d2).abs()), (m1 - m2).abs()))
    for k in range(3):
        chk("head", f"output {k}", o[k].detach(), want[k], 1e-4 if dtype == "fp32" else 2e-2)
    print(f"SegCD-{encoder} {dtype} B={B} {H}x{W} layer-local worst relative l2: " + ", ".join(f"{k} {v[0]:.1e} ({v[1]})" for k, v in worst.items()))


@pytest.mark.parametrize("fixture,classes,encoder", FIXTURES[:4])
def test_segcd_bf16_tracks_reference_vectors(golden, fixture, classes, encoder):
    """bf16 storage end to end against the reference's vectors (G10): eval-mode maps at 4e-2 relative l2 (measured 2.1e-2: one
    bf16 rounding per stored activation over ~110 layers), training loss at 2e-2; gradients: the head's and the last decoder
    block's (a few layers from the loss) keep their direction, every tensor keeps its magnitude -- deeper directions are
    not comparable on this synthetic network even between two fp32 evaluation orders (see the layer-local test above, which
    bounds every kernel instead)."""
    g = golden(fixture)
    seed = int(g["seed"])
    x1, x2 = t(g["x1"]).to(DEV), t(g["x2"]).to(DEV)
    m = SegCD(encoder_name=encoder, classes=classes, dtype="bf16")
    m.load_state_dict(G.synth_state(3, classes, seed, perturb_running=True, encoder=encoder))
    m.to(DEV).eval()
    with torch.no_grad():
        o = m(x1, x2)
    for k, v in zip(("m1", "m2"), o):
        r, _ = rel_l2_cos(v.cpu().numpy(), g[f"eval/{k}"])
        assert r <= 4e-2, (k, r)
    m = SegCD(encoder_name=encoder, classes=classes, dtype="bf16")
    m.load_state_dict(G.synth_state(3, classes, seed, encoder=encoder))
    m.to(DEV).train()
    m1, m2, ch = m(x1, x2)
    loss = _loss(m1, m2, ch, t(g["seg_target"]).to(DEV), t(g["target"]).to(DEV))
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))
    loss.backward()
    from tests._util import gf_index
    ratios = []
    for name, p in m.named_parameters():
        ref = g["gf/" + name]
        got = p.grad.detach().cpu().numpy().ravel()[gf_index(name, p.numel())]
        nr = float(np.linalg.norm(ref))
        if nr < 1e-10:
            continue
        ratios.append(float(np.linalg.norm(got)) / nr)
        if name.startswith("segmentation_head") or name.startswith("decoder.blocks.4.conv2"):
            r, c = rel_l2_cos(got, ref)
            assert c >= (0.99 if name.startswith("segmentation_head") else 0.9), (name, r, c)
    ratios = np.array(ratios)
    print(f"SegCD bf16 vs reference {fixture[:3].upper()}: gradient norm ratios median {np.median(ratios):.3f}, range [{ratios.min():.2f}, {ratios.max():.2f}]")
    assert 0.8 <= np.median(ratios) <= 1.25 and ratios.min() >= 0.4 and ratios.max() <= 2.5


def test_segcd_bf16_training_tracks_fp32():
    """The same 30 Adam steps on a fixed synthetic set in fp32 and in bf16: both must learn (loss falls by > 25 %), and the bf16
    end of the curve (mean of the last 10 steps) must lie within a factor of two of fp32's.  The band is wide on purpose: three
    fp32 runs of this loop that differ only in summation order ended at 0.233, 0.271 and 0.384 (30 Adam steps at lr 1e-3 on a
    freshly initialised ResNet-50 amplify last-bit differences); kernel-level agreement is what the layer-local test bounds."""
    from stcd_amd import synth
    from stcd_amd.optim import FlatAdamW
    a, b, lab = synth.make_batch(8, 64, 64, seed=3)
    A, Bt, L = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), torch.from_numpy(lab).to(DEV).float().unsqueeze(1)
    curves = {}
    for dt in ("fp32", "bf16"):
        torch.manual_seed(5)
        m = SegCD(encoder_name="resnet50", dtype=dt).to(DEV).train()
        opt = FlatAdamW(m, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
        losses = []
        for _ in range(30):
            opt.zero_grad(set_to_none=True)
            loss = bce_dice_with_logits(m(A, Bt)[2], L)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        curves[dt] = np.array(losses)
        assert np.isfinite(curves[dt]).all()
        assert curves[dt][-5:].mean() < 0.75 * curves[dt][:3].mean(), (dt, curves[dt])
    tail32, tail16 = curves["fp32"][-10:].mean(), curves["bf16"][-10:].mean()
    print(f"SegCD 30 steps: fp32 {curves['fp32'][0]:.3f} -> {tail32:.3f}, bf16 {curves['bf16'][0]:.3f} -> {tail16:.3f}")
    assert 0.5 * tail32 <= tail16 <= 2.0 * tail32


def test_semi_supervised_step_of_train_stcd_vs_oracle():
    """The step of the semi-supervised script (/root/reference/train_stcd.py:421-447): labelled and pseudo-change pairs
    concatenated into ONE forward of SegCD, loss = BCE+Dice(sigmoid(mask_t1)[:B], seg label) + BCE+Dice(sigmoid(change), labels)
    + contrastive_loss(sigmoid(change), cd_label, pseudo_label) -- every piece through the engine's C ABI (fp32 mode), against
    the fp64 oracle (network, losses, and autograd through them)."""
    from oracle import fcsiam_ref as R
    from stcd_amd.losses import cd_loss, contrastive_loss
    B, H, W = 2, 64, 64
    rng = np.random.default_rng(91)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    image_A, image_B, CA, CB = f(B, 3, H, W), f(B, 3, H, W), f(B, 3, H, W), f(B, 3, H, W)
    lab = lambda p: torch.from_numpy((rng.random((B, 1, H, W)) < p).astype(np.float32))
    s_label_A, cd_label, CL = lab(0.3), lab(0.2), lab(0.25)
    st = G.synth_state(3, 1, 17)
    m = SegCD(encoder_name="resnet50", dtype="fp32")
    m.load_state_dict(st)
    m.to(DEV).train()
    dA, dB, dl = torch.cat((image_A, CA)).to(DEV), torch.cat((image_B, CB)).to(DEV), torch.cat((cd_label, CL)).to(DEV)
    seg_A, seg_B, diff = m(dA, dB)
    cp = torch.sigmoid(diff)
    parts = (cd_loss(torch.sigmoid(seg_A)[:B], s_label_A.to(DEV)), cd_loss(cp, dl), contrastive_loss(cp, cd_label.to(DEV), CL.to(DEV)))
    (parts[0] + parts[1] + parts[2]).backward()

    st64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in st.items()}
    params = [k for k, v in st64.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        st64[k].requires_grad_(True)
    r1, r2, rd = G.forward(st64, torch.cat((image_A, CA)).double(), torch.cat((image_B, CB)).double(), training=True)
    rcp = torch.sigmoid(rd)
    rparts = (R.cd_loss(torch.sigmoid(r1)[:B], s_label_A.double()), R.cd_loss(rcp, torch.cat((cd_label, CL)).double()),
              R.contrastive_loss(rcp, cd_label.double(), CL.double()))
    (rparts[0] + rparts[1] + rparts[2]).backward()
    for got, want, name in zip(parts, rparts, ("seg", "cd", "contrastive")):
        assert abs(got.item() - want.item()) <= 2e-3 * max(1.0, abs(want.item())), (name, got.item(), want.item())
    worst = (0.0, 1.0)
    for name, p in m.named_parameters():
        ref = st64[name].grad
        if ref is None or float(ref.abs().max()) < 1e-12:
            continue
        r, c = rel_l2_cos(p.grad.cpu().double().numpy(), ref.numpy())
        worst = (max(worst[0], r), min(worst[1], c))
    print(f"train_stcd step, fp32 engine vs fp64 oracle: losses {[round(p.item(), 5) for p in parts]}, worst gradient rel-l2 {worst[0]:.2e} / cos {worst[1]:.6f}")
    assert worst[0] <= SEG_REL and worst[1] >= SEG_COS, worst


def test_segcd_full_size_properties_bf16():
    """SegCD at the bench's size (16 pairs of 256x256, bf16), where the CPU oracle would take minutes, through size-independent
    properties: (a) eval mode treats pairs independently -- the 16-pair batch equals its two 8-pair halves bit for bit for all
    three maps (every GEMM / conv tile walk, the stem, the decoder's concat slices); (b) the backward is linear in the output
    gradients: doubling them doubles every parameter gradient (up to the order of the atomically folded slab sums); (c) the
    `change` map obeys its definition min(., |mask_t1 - mask_t2|) <= |mask_t1 - mask_t2|; (d) everything finite, every parameter
    receives a gradient; (e) the same step twice gives bit-identical gradients."""
    from stcd_amd import synth
    a, b, _ = synth.make_batch(16, 256, 256, seed=79)
    A, Bt = t(a).to(DEV), t(b).to(DEV)
    torch.manual_seed(8)
    m = SegCD(encoder_name="resnet50", dtype="bf16").to(DEV)
    m.eval()
    with torch.no_grad():
        full = [o.clone() for o in m(A, Bt)]
        h0 = [o.clone() for o in m(A[:8], Bt[:8])]
        h1 = [o.clone() for o in m(A[8:], Bt[8:])]
    for k in range(3):
        assert torch.isfinite(full[k]).all()
        assert torch.equal(full[k], torch.cat([h0[k], h1[k]])), k
    assert (full[2] <= (full[0] - full[1]).abs() + 1e-6).all()
    m.train()
    grads = []
    for scale in (1.0, 2.0, 1.0):
        m.zero_grad(set_to_none=True)
        m._steps = 0                                     # the same step counter every pass
        o = m(A, Bt)
        torch.autograd.backward(o, [torch.ones_like(x) * 1e-3 * scale * w for x, w in zip(o, (1.0, -0.5, 2.0))])
        grads.append(torch.cat([p.grad.flatten() for p in m.parameters()]).clone())
        if scale == 1.0 and len(grads) == 1:
            for name, p in m.named_parameters():
                assert torch.isfinite(p.grad).all(), name
                assert p.grad.abs().max().item() > 0, name
    rel = ((grads[1] - 2.0 * grads[0]).abs().max() / grads[1].abs().max()).item()
    assert rel < 5e-5, rel
    # (e) run-to-run reproducibility (round 4: every reduction of the engine is fixed-order): the same step twice, bit for bit
    assert torch.equal(grads[0], grads[2]), float((grads[0] - grads[2]).abs().max())


# ---------------------------------------------------------------------------------------------------------------------
# UnetSeg: the single-image twin (decoders/unet/model.py:109-171) train_sup.py:303 trains
@pytest.mark.parametrize("tag,encoder,classes", [("r50", "resnet50", 1), ("r34", "resnet34", 2)])
def test_unetseg_fp32_matches_reference_vectors(golden, tag, encoder, classes):
    g = golden("g15_unetseg.npz")
    seed = int(g[f"{tag}/seed"])
    x = t(g[f"{tag}/x"]).to(DEV)
    m = UnetSeg(encoder_name=encoder, classes=classes, dtype="fp32")
    m.load_state_dict(G.synth_state(3, classes, seed, perturb_running=True, encoder=encoder))
    m.to(DEV).eval()
    with torch.no_grad():
        o = m(x)
    assert tuple(o.shape) == tuple(g[f"{tag}/eval"].shape)
    np.testing.assert_allclose(o.cpu().numpy(), g[f"{tag}/eval"], rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(g[f"{tag}/eval"]).max())))
    m = UnetSeg(encoder_name=encoder, classes=classes, dtype="fp32")
    m.load_state_dict(G.synth_state(3, classes, seed, encoder=encoder))
    m.to(DEV).train()
    out = m(x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"{tag}/train"], rtol=2e-3, atol=2e-3)
    loss = bce_dice_with_logits(out, t(g[f"{tag}/target"]).to(DEV))        # train_sup.py:134-135: sigmoid + criterion (= cd_loss)
    assert abs(loss.item() - float(g[f"{tag}/loss"])) < 2e-4
    loss.backward()
    for name, p in m.named_parameters():
        if float(np.abs(g[f"{tag}/gs/" + name][1])) < 1e-12:
            continue
        check_grad(name, p.grad, g, rel_max=SEG_REL, cos_min=SEG_COS, prefix=tag + "/", tag=f"fp32 UnetSeg-{encoder} vs reference G15")
    sd = m.state_dict()
    for k in [k for k in g if k.startswith(tag + "/rs/")]:
        np.testing.assert_allclose(sd[k[len(tag) + 4:]].cpu().numpy(), g[k], rtol=1e-4, atol=5e-5, err_msg=k)      # one BatchNorm call per forward


def test_unetseg_bf16_and_checkpoint_exchange_with_segcd(golden):
    """bf16 UnetSeg against G15 (eval maps, training loss, head / last-decoder gradient directions), a few Adam steps, and the
    supervised -> change-detection hand-over of the paper's pipeline: UnetSeg's state_dict loads into SegCD unchanged and
    SegCD's mask_t1 in eval mode IS UnetSeg's output."""
    from stcd_amd.optim import FlatAdam
    from tests._util import gf_index
    g = golden("g15_unetseg.npz")
    seed = int(g["r50/seed"])
    x = t(g["r50/x"]).to(DEV)
    m = UnetSeg(encoder_name="resnet50", dtype="bf16")
    m.load_state_dict(G.synth_state(3, 1, seed, perturb_running=True))
    m.to(DEV).eval()
    with torch.no_grad():
        o = m(x)
        r, _ = rel_l2_cos(o.cpu().numpy(), g["r50/eval"])
        assert r <= 4e-2, r
        cd = SegCD(encoder_name="resnet50", dtype="bf16")
        cd.load_state_dict(m.state_dict())
        cd.to(DEV).eval()
        m1, m2, ch = cd(x, x.flip(0))
        assert torch.equal(m1, o) and torch.equal(m2, o.flip(0))
    m = UnetSeg(encoder_name="resnet50", dtype="bf16")
    m.load_state_dict(G.synth_state(3, 1, seed))
    m.to(DEV).train()
    tgt = t(g["r50/target"]).to(DEV)
    loss = bce_dice_with_logits(m(x), tgt)
    assert abs(loss.item() - float(g["r50/loss"])) < 2e-2 * abs(float(g["r50/loss"]))
    loss.backward()
    for name, p in m.named_parameters():
        if name.startswith("segmentation_head") or name.startswith("decoder.blocks.4.conv2"):
            got = p.grad.detach().cpu().numpy().ravel()[gf_index(name, p.numel())]
            _, c = rel_l2_cos(got, g["r50/gf/" + name])
            assert c >= (0.99 if name.startswith("segmentation_head") else 0.9), (name, c)
    opt = FlatAdam(m, lr=1e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = bce_dice_with_logits(m(x), tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses


# ---------------------------------------------------------------------------------------------------------------------
# FFCTLCD (decoders/unet/model.py:335-423): the decoder also runs on |f1 - f2| -- a third BatchNorm group on the engine
@pytest.mark.parametrize("tag,encoder,classes", [("r34", "resnet34", 1), ("r50", "resnet50", 2)])
def test_ffctlcd_fp32_matches_reference_vectors(golden, tag, encoder, classes):
    from tests.test_oracle_golden import _gs_check
    g = golden("g16_ffctlcd.npz")
    seed = int(g[f"{tag}/seed"])
    x1, x2 = t(g[f"{tag}/x1"]).to(DEV), t(g[f"{tag}/x2"]).to(DEV)
    m = FFCTLCD(encoder_name=encoder, classes=classes, dtype="fp32")
    m.load_state_dict(G.synth_state(3, classes, seed, perturb_running=True, encoder=encoder))
    m.to(DEV).eval()
    with torch.no_grad():
        o = m(x1, x2)
    for k, v in zip(("m1", "m2", "change"), o):
        ref = g[f"{tag}/eval/{k}"]
        np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    m = FFCTLCD(encoder_name=encoder, classes=classes, dtype="fp32")
    m.load_state_dict(G.synth_state(3, classes, seed, encoder=encoder))
    m.to(DEV).train()
    m1, m2, ch = m(x1, x2)
    for k, v in zip(("m1", "m2", "change"), (m1, m2, ch)):
        np.testing.assert_allclose(v.detach().cpu().numpy(), g[f"{tag}/train/{k}"], rtol=2e-3, atol=2e-3, err_msg=k)
    loss = _loss(m1, m2, ch, t(g[f"{tag}/seg_target"]).to(DEV), t(g[f"{tag}/target"]).to(DEV))
    assert abs(loss.item() - float(g[f"{tag}/loss"])) < 2e-4
    loss.backward()
    for name, p in m.named_parameters():
        if float(np.abs(g[f"{tag}/gs/" + name][1])) < 1e-12:
            continue
        if f"{tag}/gf/{name}" in g:
            check_grad(name, p.grad, g, rel_max=SEG_REL, cos_min=SEG_COS, prefix=tag + "/", tag=f"fp32 FFCTLCD-{encoder} vs reference G16")
        else:
            _gs_check(name, p.grad.cpu(), g[f"{tag}/gs/" + name], SEG_REL)
    sd = m.state_dict()
    for k in [k for k in g if k.startswith(tag + "/rs/")]:      # decoder BatchNorms: |f1 - f2| first, then date 1, date 2; 3 calls
        np.testing.assert_allclose(sd[k[len(tag) + 4:]].cpu().numpy(), g[k], rtol=1e-4, atol=5e-5, err_msg=k)


@pytest.mark.parametrize("dtype,encoder,B,H,W", [("fp32", "resnet18", 2, 64, 64), ("bf16", "resnet34", 2, 64, 96), ("bf16", "resnet50", 3, 96, 64)])
def test_ffctlcd_layer_local_in_place(dtype, encoder, B, H, W):
    """The three-group decoder checked in place (stcd_ws_tensor_*): every decoder layer's conv output / weight gradient /
    per-group BatchNorm + ReLU from its own stored tensors, the |f1 - f2| group of every decoder input, the skip / up-sampling
    plumbing with three groups, and the head."""
    rng = np.random.default_rng(51)
    x1 = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
    x2 = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
    tgt = torch.from_numpy((rng.random((B, 1, H, W)) < 0.2).astype(np.float32)).to(DEV)
    m = FFCTLCD(encoder_name=encoder, dtype=dtype)
    m.load_state_dict(G.synth_state(3, 1, 13, encoder=encoder))
    m._engine.set_debug(1)
    m.to(DEV).train()
    o = m(x1, x2)
    loss = bce_dice_with_logits(o[2], tgt) + bce_dice_with_logits(o[0], tgt) + 0.5 * o[1].mean()
    loss.backward()
    torch.cuda.synchronize()
    ws = m._engine.ws_tensors()
    tol = 2e-4 if dtype == "fp32" else 6e-3
    wq = (lambda w: w.detach().to(torch.bfloat16).float()) if dtype == "bf16" else (lambda w: w.detach())
    worst = {}

    def chk(kind, name, got, want, t_=tol):
        r = float((got.float() - want).norm() / want.norm().clamp_min(1e-30))
        if r >= worst.get(kind, (0.0, ""))[0]:
            worst[kind] = (r, name)
        assert r <= t_, (kind, name, r)

    basic = G.ENCODERS[encoder][0] == 1
    last = {li: f"encoder.layer{li}.{G.ENCODERS[encoder][1][li - 1] - 1}.conv{2 if basic else 3}" for li in (1, 2, 3, 4)}
    skips = [last[3], last[2], last[1], "encoder.conv1"]
    rq = (lambda v: v.to(torch.bfloat16).float()) if dtype == "bf16" else (lambda v: v)      # the difference is stored in the activation type
    x = _nchw(ws[last[4] + ".A"])                                    # [2B]: the two dates
    x = torch.cat([x, rq((x[:B] - x[B:]).abs())])                    # + |f5 - f5|
    for i, blk in enumerate(m.decoder.blocks):
        cat = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
        if i < 4:
            sk = _nchw(ws[skips[i] + ".A"])
            cat = torch.cat([cat, torch.cat([sk, rq((sk[:B] - sk[B:]).abs())])], dim=1)
        stored = _nchw(ws[f"decoder.blocks.{i}.conv1.0.in"])
        assert stored.shape[0] == 3 * B
        assert torch.equal(stored, cat), i                         # three groups: date 1, date 2, |difference|
        for cname, (conv, bn) in ((f"decoder.blocks.{i}.conv1.0", (blk.conv1[0], blk.conv1[1])), (f"decoder.blocks.{i}.conv2.0", (blk.conv2[0], blk.conv2[1]))):
            X, Y, A, dY = _nchw(ws[cname + ".in"]), _nchw(ws[cname + ".Y"]), _nchw(ws[cname + ".A"]), _nchw(ws[cname + ".dY"])
            Wt = wq(conv.weight)
            chk("conv output", cname, Y, torch.nn.functional.conv2d(X, Wt, None, 1, 1))
            chk("weight gradient", cname, conv.weight.grad, torch.nn.grad.conv2d_weight(X, conv.weight.shape, dY, 1, 1), 2e-4 if dtype == "fp32" else 2e-3)
            chk("input gradient", cname, _nchw(ws[cname + ".dIn"]), torch.nn.grad.conv2d_input(X.shape, Wt, dY, 1, 1))
            outs = []
            for d in range(3):
                y = Y[d * B:(d + 1) * B]
                mu, var = y.mean(dim=(0, 2, 3), keepdim=True), y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
                outs.append(torch.relu((y - mu) * torch.rsqrt(var + 1e-5) * bn.weight.detach().view(1, -1, 1, 1) + bn.bias.detach().view(1, -1, 1, 1)))
            chk("bn+relu", cname, A, torch.cat(outs), 1e-4 if dtype == "fp32" else 1.2e-2)
        x = _nchw(ws[f"decoder.blocks.{i}.conv2.0.A"])
    hw, hb = wq(m.segmentation_head[0].weight), m.segmentation_head[0].bias.detach()
    head = lambda t_: torch.nn.functional.conv2d(t_, hw, hb, padding=1)
    m1, m2, dif = head(x[:B]), head(x[B:2 * B]), head(x[2 * B:])
    want = (m1, m2, torch.min(dif, (m1 - m2).abs()))
    for k in range(3):
        chk("head", f"output {k}", o[k].detach(), want[k], 1e-4 if dtype == "fp32" else 2e-2)
    print(f"FFCTLCD-{encoder} {dtype} B={B} {H}x{W} layer-local worst relative l2: " + ", ".join(f"{k} {v[0]:.1e} ({v[1]})" for k, v in worst.items()))


def test_ffctlcd_bf16_tracks_reference_vectors_and_trains(golden):
    """bf16 FFCTLCD against G16 (eval maps at 4e-2 relative l2, training loss at 2e-2, the head's gradient direction), then 12 Adam
    steps on the fixture's batch: finite, and the loss falls."""
    from stcd_amd.optim import FlatAdam
    from tests._util import gf_index
    g = golden("g16_ffctlcd.npz")
    seed = int(g["r34/seed"])
    x1, x2 = t(g["r34/x1"]).to(DEV), t(g["r34/x2"]).to(DEV)
    m = FFCTLCD(encoder_name="resnet34", dtype="bf16")
    m.load_state_dict(G.synth_state(3, 1, seed, perturb_running=True, encoder="resnet34"))
    m.to(DEV).eval()
    with torch.no_grad():
        o = m(x1, x2)
    for k, v in zip(("m1", "m2"), o):
        r, _ = rel_l2_cos(v.cpu().numpy(), g[f"r34/eval/{k}"])
        assert r <= 4e-2, (k, r)
    m = FFCTLCD(encoder_name="resnet34", dtype="bf16")
    m.load_state_dict(G.synth_state(3, 1, seed, encoder="resnet34"))
    m.to(DEV).train()
    seg, tgt = t(g["r34/seg_target"]).to(DEV), t(g["r34/target"]).to(DEV)
    loss = _loss(*m(x1, x2), seg, tgt)
    assert abs(loss.item() - float(g["r34/loss"])) < 2e-2 * abs(float(g["r34/loss"]))
    loss.backward()
    for name, p in m.named_parameters():
        if name.startswith("segmentation_head"):
            got = p.grad.detach().cpu().numpy().ravel()[gf_index(name, p.numel())]
            _, c = rel_l2_cos(got, g["r34/gf/" + name])
            assert c >= 0.99, (name, c)
    opt = FlatAdam(m, lr=1e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = _loss(*m(x1, x2), seg, tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses
