"""CPU-side checks of the ChangeFormerV6 row: the oracle's pinned blocks against the vectors captured from
the reference's own classes (tests/golden/g17_cf_base.npz), the parameter table of the reference's registration order against the
engine's table and the drop-in module, the dropout-hash restatement against the library, host-side argument checking."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import changeformer_ref as R
from stcd_amd import _lib
from stcd_amd._lib import StcdError
from stcd_amd.changeformer import ChangeFormerV6

TINY = dict(embed_dims=(64, 64, 128, 128), depths=(2, 1, 1, 2), num_heads=(1, 2, 2, 4))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_pinned_decoder_head_blocks_match_the_reference_vectors(golden):
    """conv_layer / upsample_conv / residual_block == ConvLayer / UpsampleConvLayer / ResidualBlock of
    /root/reference/models/ChangeFormerBaseNetworks.py:85-120 (outputs and every gradient)."""
    g = golden("g17_cf_base.npz")
    for tag in ("res", "up", "conv"):
        x = _t(g[f"{tag}/x"]).clone().requires_grad_(True)
        p = {k[len(tag) + 3:]: _t(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith(f"{tag}/p/")}
        if tag == "res":
            y = R.residual_block(x, p["conv1.conv2d.weight"], p["conv1.conv2d.bias"], p["conv2.conv2d.weight"], p["conv2.conv2d.bias"])
        elif tag == "up":
            y = R.upsample_conv(x, p["conv2d.weight"], p["conv2d.bias"])
        else:
            y = R.conv_layer(x, p["conv2d.weight"], p["conv2d.bias"], 1)
        np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}/y"], rtol=1e-5, atol=1e-6)
        y.backward(_t(g[f"{tag}/gy"]))
        np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}/gx"], rtol=1e-4, atol=1e-5)
        for k, v in p.items():
            np.testing.assert_allclose(v.grad.numpy(), g[f"{tag}/g/{k}"], rtol=1e-4, atol=1e-4, err_msg=f"{tag} {k}")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decoder_matches_the_reference_classes(golden, tag):
    """oracle.changeformer_ref.decoder (+ _conv_diff, _make_pred) == the reference's OWN DecoderTransformer_v3 / MLP / conv_diff /
    make_prediction / resize (/root/reference/models/ChangeFormer.py:238-257,677-688,1138-1157,1475-1631), compiled from its file
    by ast (tests/golden/make_golden.py g21; timm is never touched): the five output maps in eval mode, and in training mode --
    Dropout p = 0, and with the recorded element-wise masks of the eight nn.Dropout(0.6) sites -- outputs, loss, every parameter's
    gradient, the gradients of all eight input features and the BatchNorm running statistics, at 1e-5."""
    g = golden("g21_cf_decoder.npz")
    meta = g[f"{tag}/meta"]
    B, emb, out, h, w = (int(v) for v in meta[:5])
    cfg = R.CFConfig(out_ch=out)
    feats = [[_t(g[f"{tag}/f{i + 1}_{s}"]) for s in range(4)] for i in range(2)]
    wts = [0.5, -0.25, 0.75, 1.0, 2.0]

    def state():
        st = {"TDec_x2." + k[len(tag) + 6:]: _t(v).clone() for k, v in g.items() if k.startswith(f"{tag}/init/")}
        for k, v in st.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        return st

    def close(a, b, what, rtol=1e-5, atol=1e-5):
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * max(1.0, float(np.abs(b).max())), err_msg=f"{tag} {what}")

    st = state()
    with torch.no_grad():
        outs = R.decoder(cfg, st, feats[0], feats[1], False, None)
    for i, o in enumerate(outs):
        close(o.numpy(), g[f"{tag}/eval/out{i}"], f"eval out{i}")
    for mode in ("train_p0", "train_masks"):
        st = state()
        f = [[x.clone().requires_grad_(True) for x in fs] for fs in feats]
        masks = {} if mode == "train_p0" else {k[len(tag) + 18:]: _t(v) for k, v in g.items() if k.startswith(f"{tag}/train_masks/mask/")}
        assert mode == "train_p0" or len(masks) == 8
        outs = R.decoder(cfg, st, f[0], f[1], True, masks)
        for i, o in enumerate(outs):
            close(o.detach().numpy(), g[f"{tag}/{mode}/out{i}"], f"{mode} out{i}")
        loss = sum(wt * (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum() for wt, o in zip(wts, outs)) / 100.0
        assert abs(loss.item() - float(g[f"{tag}/{mode}/loss"])) <= 1e-5 * max(1.0, abs(float(g[f"{tag}/{mode}/loss"])))
        loss.backward()
        n = 0
        for k, v in st.items():
            key = f"{tag}/{mode}/g/{k[8:]}"
            if key in g:
                close(v.grad.numpy(), g[key], f"{mode} grad {k}", rtol=1e-4, atol=2e-5)
                n += 1
            bkey = f"{tag}/{mode}/bn/{k[8:]}"
            if bkey in g:
                close(v.detach().numpy().astype(np.float64), g[bkey].astype(np.float64), f"{mode} {k}")
        assert n == sum(1 for k in g if k.startswith(f"{tag}/{mode}/g/")) and n > 60
        for i in range(2):
            for s in range(4):
                close(f[i][s].grad.numpy(), g[f"{tag}/{mode}/gf{i + 1}_{s}"], f"{mode} d feature {i + 1}.{s}", rtol=1e-4, atol=2e-5)


def test_depthwise_step_of_mix_ffn_matches_the_reference_class(golden):
    """oracle.changeformer_ref.dwconv_tokens (the depth-wise step inside mix_ffn) == the reference's DWConv
    (/root/reference/models/ChangeFormer.py:512-523, compiled from the file by ast): output and all three gradients."""
    g = golden("g21_cf_decoder.npz")
    x = _t(g["dw/x"]).clone().requires_grad_(True)
    w = _t(g["dw/w"]).clone().requires_grad_(True)
    b = _t(g["dw/b"]).clone().requires_grad_(True)
    y = R.dwconv_tokens(x, w, b, 6, 5)
    np.testing.assert_allclose(y.detach().numpy(), g["dw/y"], rtol=1e-5, atol=1e-6)
    y.backward(_t(g["dw/gy"]))
    np.testing.assert_allclose(x.grad.numpy(), g["dw/gx"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(w.grad.numpy(), g["dw/gw"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.grad.numpy(), g["dw/gb"], rtol=1e-4, atol=1e-5)


def test_parameter_table_engine_table_and_module_agree():
    """One order, three places: the oracle's table (the reference's registration order), the engine's flat layout
    (stcd_param_info) and the nn.Module's state_dict.  41 028 730 parameters: the figure the ChangeFormer paper quotes for V6."""
    m = ChangeFormerV6()
    tab = R.param_table(R.CFConfig())
    sd = m.state_dict()
    assert list(sd.keys()) == [n for n, _, _ in tab]
    for (n, shape, _), v in zip(tab, sd.values()):
        assert tuple(v.shape) == tuple(shape), n
    eng = [(p.name, p.shape) for p in m._engine.params]
    assert eng == [(n, tuple(s)) for n, s, k in tab if k not in ("rm", "rv", "nbt")]
    assert sum(p.numel() for p in m.parameters()) == 41028730
    # a reference-shaped state dict loads strictly and round-trips
    st = R.synth_state(R.CFConfig(), 9, perturb_running=True)
    m.load_state_dict(st, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, st[k]), k
    # two gradient stages for data-parallel overlap: decoder first (its parameters come last in the flat buffer)
    (b0, e0), (b1, e1) = m._engine.stage0_range, m._engine.stage1_range
    assert b1 == 0 and e1 == b0 and e0 == m._engine.param_floats
    first_dec = next(p for p in m._engine.params if p.name.startswith("TDec_x2."))
    assert first_dec.offset == b0


def test_other_configurations_of_the_class_family():
    t = ChangeFormerV6(3, 1, embed_dim=64, config=dict(TINY))
    tab = R.param_table(R.CFConfig.tiny(out_ch=1))
    assert list(t.state_dict().keys()) == [n for n, _, _ in tab]
    with pytest.raises(StcdError):
        ChangeFormerV6(embed_dim=96)                          # decoder width must be a power of two (BatchNorm kernels)
    with pytest.raises(StcdError):
        ChangeFormerV6(config=dict(embed_dims=(60, 128, 320, 512)))
    with pytest.raises(StcdError):
        ChangeFormerV6(config=dict(bogus=1))


def test_default_initialisation_follows_the_reference():
    """encoder: Linear trunc_normal(std .02) / bias 0, LayerNorm 1 / 0, Conv2d N(0, sqrt(2 / fan_out)) / bias 0
    (ChangeFormer.py:1411-1424); decoder: torch defaults (PReLU 0.25, BatchNorm 1 / 0)."""
    torch.manual_seed(0)
    m = ChangeFormerV6(embed_dim=64, config=dict(TINY))
    sd = m.state_dict()
    w = sd["Tenc_x2.block1.0.mlp.fc1.weight"]
    assert abs(float(w.std()) - 0.02) < 2e-3           # (timm's trunc_normal_ truncates at +-2 ABSOLUTE: never reached at std 0.02)
    assert float(sd["Tenc_x2.block1.0.mlp.fc1.bias"].abs().max()) == 0.0
    assert torch.equal(sd["Tenc_x2.norm1.weight"], torch.ones(64))
    dw = sd["Tenc_x2.block1.0.mlp.dwconv.dwconv.weight"]
    assert abs(float(dw.std()) - (2.0 / 9.0) ** 0.5) < 0.05                               # fan_out = 3 * 3 * C / groups = 9
    pe = sd["Tenc_x2.patch_embed1.proj.weight"]
    assert abs(float(pe.std()) - (2.0 / (49 * 64)) ** 0.5) < 3e-3
    assert float(sd["TDec_x2.diff_c4.1.weight"]) == 0.25
    assert torch.equal(sd["TDec_x2.linear_fuse.1.weight"], torch.ones(64))


def test_dropout_hash_restatement_matches_the_library():
    lib = _lib.lib()
    for seed in (0, 7, 0x5EED1234ABCD, 2 ** 64 - 1):
        for site in (0, 3, 200):
            assert lib.stcd_cf_site_seed(C.c_uint64(seed), site) == R.site_seed(seed, site)
    # keep rates of the restated hash (the engine's per-element rule): unbiased masks at the reference's rates
    for p in (0.1, 0.6):
        k = R.hash_keep(1 << 20, R.site_seed(123, 4), p)
        assert abs(k.mean() - (1 - p)) < 2e-3
    # consecutive sites are independent draws
    a, b = R.hash_keep(1 << 16, R.site_seed(5, 0), 0.5), R.hash_keep(1 << 16, R.site_seed(5, 1), 0.5)
    assert abs((a == b).mean() - 0.5) < 1e-2


def test_oracle_eval_is_deterministic_and_masks_change_training():
    cfg = R.CFConfig.tiny()
    st = R.synth_state(cfg, 3, perturb_running=True)
    g = torch.Generator().manual_seed(1)
    x1, x2 = torch.randn(1, 3, 32, 32, generator=g), torch.randn(1, 3, 32, 32, generator=g)
    with torch.no_grad():
        a = R.forward(cfg, st, x1, x2, False)
        b = R.forward(cfg, st, x1, x2, False)
        assert all(torch.equal(u, v) for u, v in zip(a, b))
        assert [tuple(o.shape) for o in a] == [(1, 2, 1, 1), (1, 2, 2, 2), (1, 2, 4, 4), (1, 2, 8, 8), (1, 2, 32, 32)]
        m1 = R.engine_masks(cfg, 1, 32, 32, 11)
        m2 = R.engine_masks(cfg, 1, 32, 32, 12)
        t1 = R.forward(cfg, {k: v.clone() for k, v in st.items()}, x1, x2, True, m1)[-1]
        t2 = R.forward(cfg, {k: v.clone() for k, v in st.items()}, x1, x2, True, m2)[-1]
        assert not torch.allclose(t1, t2)


def test_cpu_tensors_and_unsupported_uses_are_refused():
    m = ChangeFormerV6(embed_dim=64, config=dict(TINY))
    with pytest.raises(StcdError):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))
    with pytest.raises(StcdError):
        m.set_dropout_masks({})
    with pytest.raises(StcdError):
        m.set_dropout_p(0.5)


def test_define_G_builds_the_engine_module():
    from types import SimpleNamespace

    from stcd_amd.networks import define_G
    net = define_G(SimpleNamespace(net_G="ChangeFormerV6", n_class=2, embed_dim=64))
    assert isinstance(net, ChangeFormerV6) and net.embedding_dim == 64
    # init_weights('normal', 0.02) reached the Linear / Conv holders and the BatchNorm gains (networks.py:85-116)
    sd = net.state_dict()
    assert abs(float(sd["TDec_x2.dense_1x.0.conv1.conv2d.weight"].std()) - 0.02) < 2e-3
    assert abs(float(sd["TDec_x2.linear_fuse.1.weight"].mean()) - 1.0) < 2e-2 and float(sd["TDec_x2.linear_fuse.1.weight"].std()) > 1e-3
