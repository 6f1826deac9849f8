"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch) of the reference's ChangeFormerV6 (SURVEY.md section 8 row f-4,
BASELINE.json configs[4]).  Never imported by the product (stcd_amd/, bench.py's timed region).

Restated from text; every function cites the lines it follows under /root/reference/models/:
  OverlapPatchEmbed            ChangeFormer.py:195-236   (Conv2d k, stride, pad k//2  ->  flatten  ->  LayerNorm eps 1e-5)
  Attention (SR-attention)     ChangeFormer.py:298-358   (q / kv Linear, sr Conv2d k=s=sr + LayerNorm eps 1e-5, softmax, attn_drop,
                                                          proj, proj_drop)
  Mlp + DWConv (Mix-FFN)       ChangeFormer.py:260-295, 512-523   (fc1 -> depthwise 3x3 -> GELU -> drop -> fc2 -> drop)
  Block                        ChangeFormer.py:472-509   (x + drop_path(attn(norm1 x)); x + drop_path(mlp(norm2 x)); LN eps 1e-6)
  EncoderTransformer_v3        ChangeFormer.py:1342-1473 (4 stages; stage norm eps 1e-6; dpr = linspace(0, rate, sum(depths)))
  MLP, conv_diff, make_prediction   ChangeFormer.py:677-688, 1138-1157
  DecoderTransformer_v3        ChangeFormer.py:1475-1631
  ChangeFormerV6               ChangeFormer.py:1669-1701 (widths [64,128,320,512], depths [3,3,4,3], heads [1,2,4,8], sr [8,4,2,1],
                                                          patch 7 for ALL four embeddings (k7 s4 p3, then k7 s2 p3), drop 0.1 /
                                                          attn_drop 0.1 / drop_path 0.1, embedding_dim 256; SURVEY R10: these are
                                                          the in-tree widths, not MiT-B0)
  ConvLayer / UpsampleConvLayer / ResidualBlock   ChangeFormerBaseNetworks.py:85-120

PIN STATUS.  `models/ChangeFormer.py` cannot be imported in the authoring container (it imports `timm` at :10-11, absent).
  * DECODER: PINNED (round 4).  `resize`, `MLP`, `conv_diff`, `make_prediction`, `DecoderTransformer_v3` and `DWConv` use none of
    timm's names; tests/golden/make_golden.py (g21) compiles exactly those definitions from the reference's file by `ast` (the
    recipe of G8; with the importable models.ChangeFormerBaseNetworks classes in the namespace) and stores outputs, every
    gradient and the BatchNorm statistics of two feature pyramids (eval, train with Dropout p = 0, train with recorded masks):
    `decoder`, `_conv_diff`, `_make_pred` and `dwconv_tokens` below are checked against tests/golden/g21_cf_decoder.npz at 1e-5
    (tests/test_changeformer_cpu.py).  The three classes of `models/ChangeFormerBaseNetworks.py` are pinned by g17 as before.
  * ENCODER: **parity unpinned**.  `OverlapPatchEmbed`, `Attention`, `Mlp`, `Block`, `EncoderTransformer_v3` call timm's
    `trunc_normal_` / `to_2tuple` / `DropPath` at construction: they are restated from the text only, and what timm supplies is
    restated from its published definitions -- `DropPath` (per-sample Bernoulli(keep) mask divided by keep),
    `trunc_normal_(std=.02)` (torch.nn.init.trunc_normal_ is the same algorithm), `to_2tuple`.  No timm name is defined anywhere
    in this repository.

Randomness is explicit: every Dropout / DropPath site takes its mask from `masks` (name -> tensor already divided by keep),
so the HIP engine and this file can be run on identical masks (`engine_masks` reproduces the engine's counter hash).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_MOMENTUM, BN_EPS = 0.1, 1e-5


@dataclass
class CFConfig:
    """ChangeFormerV6.__init__ (ChangeFormer.py:1671-1691); `tiny()` keeps every code path at a size tests finish in seconds."""
    in_ch: int = 3
    out_ch: int = 2
    embed_dims: Tuple[int, ...] = (64, 128, 320, 512)
    depths: Tuple[int, ...] = (3, 3, 4, 3)
    num_heads: Tuple[int, ...] = (1, 2, 4, 8)
    sr_ratios: Tuple[int, ...] = (8, 4, 2, 1)
    mlp_ratio: int = 4
    embedding_dim: int = 256
    patch1: int = 7            # patch_embed1: k7 s4 (ChangeFormer.py:1353)
    patch: int = 7             # patch_embed2..4: k = patch_size (7 in V6, ChangeFormer.py:1682), s2
    drop_rate: float = 0.1
    attn_drop: float = 0.1
    drop_path_rate: float = 0.1
    diff_drop: float = 0.6     # conv_diff's nn.Dropout(p=0.6) (ChangeFormer.py:1143,1147)

    @staticmethod
    def tiny(out_ch: int = 2):
        return CFConfig(out_ch=out_ch, embed_dims=(64, 64, 128, 128), depths=(2, 1, 1, 2), num_heads=(1, 2, 2, 4),
                        embedding_dim=64)

    def dpr(self) -> List[float]:
        n = sum(self.depths)
        return [float(x) for x in torch.linspace(0, self.drop_path_rate, n)]


# ------------------------------------------------------------------------------------------------ parameter table
def param_table(cfg: CFConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(state_dict key, shape, kind)] in the reference's registration order (EncoderTransformer_v3.__init__ :1354-1401:
    patch_embed1..4, then block_k / norm_k per stage; Block :477-488; Attention :308-317; Mlp :265-269;
    DecoderTransformer_v3.__init__ :1498-1531).  kind: how synth_state / the default initialisation fills it."""
    t: List[Tuple[str, Tuple[int, ...], str]] = []
    E, D = cfg.embed_dims, cfg.embedding_dim

    def ln(name, c):
        t.append((name + ".weight", (c,), "ln_w"))
        t.append((name + ".bias", (c,), "ln_b"))

    def lin(name, cin, cout):
        t.append((name + ".weight", (cout, cin), "linear"))
        t.append((name + ".bias", (cout,), "bias"))

    def conv(name, cin, cout, k, groups=1, kind="conv"):
        t.append((name + ".weight", (cout, cin // groups, k, k), kind))
        t.append((name + ".bias", (cout,), "bias"))

    def bn(name, c):
        t.append((name + ".weight", (c,), "bn_w"))
        t.append((name + ".bias", (c,), "bn_b"))
        t.append((name + ".running_mean", (c,), "rm"))
        t.append((name + ".running_var", (c,), "rv"))
        t.append((name + ".num_batches_tracked", (), "nbt"))

    pre = "Tenc_x2."
    cin = cfg.in_ch
    for s in range(4):
        k = cfg.patch1 if s == 0 else cfg.patch
        conv(pre + f"patch_embed{s + 1}.proj", cin, E[s], k)
        ln(pre + f"patch_embed{s + 1}.norm", E[s])
        cin = E[s]
    for s in range(4):
        C, sr = E[s], cfg.sr_ratios[s]
        for i in range(cfg.depths[s]):
            b = pre + f"block{s + 1}.{i}."
            ln(b + "norm1", C)
            lin(b + "attn.q", C, C)
            lin(b + "attn.kv", C, 2 * C)
            lin(b + "attn.proj", C, C)
            if sr > 1:
                conv(b + "attn.sr", C, C, sr)
                ln(b + "attn.norm", C)
            ln(b + "norm2", C)
            lin(b + "mlp.fc1", C, cfg.mlp_ratio * C)
            conv(b + "mlp.dwconv.dwconv", cfg.mlp_ratio * C, cfg.mlp_ratio * C, 3, groups=cfg.mlp_ratio * C)
            lin(b + "mlp.fc2", cfg.mlp_ratio * C, C)
        ln(pre + f"norm{s + 1}", C)
    d = "TDec_x2."
    for s in (4, 3, 2, 1):
        lin(d + f"linear_c{s}.proj", E[s - 1], D)
    for s in (4, 3, 2, 1):
        conv(d + f"diff_c{s}.0", 2 * D, D, 3, kind="dconv")
        t.append((d + f"diff_c{s}.1.weight", (1,), "prelu"))
        bn(d + f"diff_c{s}.2", D)
        conv(d + f"diff_c{s}.4", D, D, 3, kind="dconv")
        t.append((d + f"diff_c{s}.5.weight", (1,), "prelu"))
        bn(d + f"diff_c{s}.6", D)
    for s in (4, 3, 2, 1):
        conv(d + f"make_pred_c{s}.0", D, cfg.out_ch, 3, kind="dconv")
        bn(d + f"make_pred_c{s}.2", cfg.out_ch)
        conv(d + f"make_pred_c{s}.3", cfg.out_ch, cfg.out_ch, 3, kind="dconv")
    conv(d + "linear_fuse.0", 4 * D, D, 1, kind="dconv")
    bn(d + "linear_fuse.1", D)
    t.append((d + "convd2x.conv2d.weight", (D, D, 4, 4), "dconvT"))
    t.append((d + "convd2x.conv2d.bias", (D,), "bias"))
    conv(d + "dense_2x.0.conv1.conv2d", D, D, 3, kind="dconv")
    conv(d + "dense_2x.0.conv2.conv2d", D, D, 3, kind="dconv")
    t.append((d + "convd1x.conv2d.weight", (D, D, 4, 4), "dconvT"))
    t.append((d + "convd1x.conv2d.bias", (D,), "bias"))
    conv(d + "dense_1x.0.conv1.conv2d", D, D, 3, kind="dconv")
    conv(d + "dense_1x.0.conv2.conv2d", D, D, 3, kind="dconv")
    conv(d + "change_probability.conv2d", D, cfg.out_ch, 3, kind="dconv")
    return t


def synth_state(cfg: CFConfig, seed: int, perturb_running: bool = False, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic weights with realistic scales (numpy PCG64): every tensor is non-trivial so that no gradient path is
    hidden by a zero bias or a unit gain; fan-in scaled so activations stay O(1) through the depth."""
    rng = np.random.default_rng(seed)
    st: Dict[str, torch.Tensor] = {}
    for name, shape, kind in param_table(cfg):
        if kind == "nbt":
            st[name] = torch.zeros((), dtype=torch.int64)
            continue
        if kind in ("linear",):
            v = rng.standard_normal(shape) * (1.0 / math.sqrt(shape[1]))
        elif kind in ("conv", "dconv"):
            fan_in = shape[1] * shape[2] * shape[3]
            v = rng.standard_normal(shape) * (1.0 / math.sqrt(fan_in))
        elif kind == "dconvT":
            fan_in = shape[0] * 4          # a k4 s2 transposed conv adds 2x2 taps per output pixel
            v = rng.standard_normal(shape) * (1.0 / math.sqrt(fan_in))
        elif kind in ("ln_w", "bn_w"):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind in ("ln_b", "bn_b", "bias"):
            v = 0.1 * rng.standard_normal(shape)
        elif kind == "prelu":
            v = np.full(shape, 0.25) + 0.05 * rng.standard_normal(shape)
        elif kind == "rm":
            v = 0.1 * rng.standard_normal(shape) if perturb_running else np.zeros(shape)
        elif kind == "rv":
            v = 1.0 + 0.2 * rng.random(shape) if perturb_running else np.ones(shape)
        else:
            raise KeyError(kind)
        st[name] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(dtype)
    return st


def default_init_state(cfg: CFConfig, seed: int) -> Dict[str, torch.Tensor]:
    """The reference's own initialisation: encoder `_init_weights` (ChangeFormer.py:1411-1424: Linear trunc_normal(std .02) /
    bias 0, LayerNorm 1 / 0, Conv2d N(0, sqrt(2 / fan_out)) / bias 0); decoder: torch defaults (kaiming_uniform(a=sqrt 5) for
    Conv / Linear weights, U(+-1/sqrt(fan_in)) biases, PReLU 0.25, BatchNorm 1 / 0)."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    for name, shape, kind in param_table(cfg):
        if kind == "nbt":
            st[name] = torch.zeros((), dtype=torch.int64)
        elif kind == "linear" and name.startswith("Tenc_x2."):
            w = torch.empty(shape)
            torch.nn.init.trunc_normal_(w, std=0.02, generator=g)
            st[name] = w
        elif kind == "conv":
            fan_out = shape[0] * shape[2] * shape[3] // (1 if shape[1] > 1 or "dwconv" not in name else shape[0])
            st[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif kind in ("linear", "dconv", "dconvT"):
            fan_in = (shape[1] if kind != "dconvT" else shape[0]) * (shape[2] * shape[3] if len(shape) == 4 else 1)
            bound = 1.0 / math.sqrt(fan_in)        # kaiming_uniform(a = sqrt(5)) == U(+-1/sqrt(fan_in))
            st[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "bias":
            if name.startswith("Tenc_x2."):
                st[name] = torch.zeros(shape)
            else:
                wshape = st[name[:-4] + "weight"].shape
                fan_in = (wshape[1] if "convd" not in name.split(".")[1] else wshape[0]) * (wshape[2] * wshape[3] if len(wshape) == 4 else 1)
                st[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        elif kind in ("ln_w", "bn_w", "rv"):
            st[name] = torch.ones(shape)
        elif kind in ("ln_b", "bn_b", "rm"):
            st[name] = torch.zeros(shape)
        elif kind == "prelu":
            st[name] = torch.full(shape, 0.25)
    return st


# ------------------------------------------------------------------------------------------------ pinned leaf blocks
def conv_layer(x, w, b, padding):
    """ConvLayer.forward (ChangeFormerBaseNetworks.py:85-96): plain Conv2d."""
    return F.conv2d(x, w, b, stride=1, padding=padding)


def upsample_conv(x, w, b):
    """UpsampleConvLayer.forward (ChangeFormerBaseNetworks.py:99-106): ConvTranspose2d(k4, stride 2, padding 1)."""
    return F.conv_transpose2d(x, w, b, stride=2, padding=1)


def residual_block(x, w1, b1, w2, b2):
    """ResidualBlock.forward (ChangeFormerBaseNetworks.py:109-120): conv2(relu(conv1 x)) * 0.1 + x."""
    out = F.relu(conv_layer(x, w1, b1, 1))
    out = conv_layer(out, w2, b2, 1) * 0.1
    return out + x


# ------------------------------------------------------------------------------------------------ restated (unpinned) parts
TAPS: Optional[Dict[str, torch.Tensor]] = None      # set to a dict to record intermediates (name -> detached tensor, the engine's
                                                    # introspection names: tests compare the engine's stored tensors in place)


def _tap(name, x, tokens_hw=None):
    """record x as NHWC [n, h, w, c]: x is NCHW, or tokens [n, N, c] with tokens_hw = (h, w)"""
    if TAPS is None:
        return
    if tokens_hw is not None:
        t = x.detach().reshape(x.shape[0], tokens_hw[0], tokens_hw[1], x.shape[2])
    else:
        t = x.detach().permute(0, 2, 3, 1)
    prev = TAPS.get(name)
    TAPS[name] = t.clone() if prev is None else torch.cat((prev, t), 0)       # the second date stacks behind the first


def _m(masks, name, x):
    """x * mask for a Dropout / DropPath site (mask already holds 1/keep); masks None: evaluation mode (identity)."""
    if masks is None:
        return x
    m = masks.get(name)
    return x if m is None else x * m.to(x.dtype)


def _bn(st, name, x, training):
    w, b = st[name + ".weight"], st[name + ".bias"]
    rm, rv = st[name + ".running_mean"], st[name + ".running_var"]
    if not training:
        return F.batch_norm(x, rm.to(x.dtype), rv.to(x.dtype), w, b, False, 0.0, BN_EPS)
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    n = x.numel() / x.shape[1]
    with torch.no_grad():
        rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(rm.dtype))
        rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * (var.detach() * n / max(n - 1, 1)).to(rv.dtype))
        st[name + ".num_batches_tracked"] += 1
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def patch_embed(st, name, x, k, stride):
    """OverlapPatchEmbed.forward (ChangeFormer.py:228-236)."""
    x = F.conv2d(x, st[name + ".proj.weight"], st[name + ".proj.bias"], stride=stride, padding=k // 2)
    _, C, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (C,), st[name + ".norm.weight"], st[name + ".norm.bias"], 1e-5)
    return x, H, W


def attention(st, name, x, H, W, heads, sr, masks):
    """Attention.forward (ChangeFormer.py:334-358)."""
    B, N, C = x.shape
    d = C // heads
    q = F.linear(x, st[name + ".q.weight"], st[name + ".q.bias"]).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = F.conv2d(x_, st[name + ".sr.weight"], st[name + ".sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x_ = F.layer_norm(x_, (C,), st[name + ".norm.weight"], st[name + ".norm.bias"], 1e-5)
    else:
        x_ = x
    kv = F.linear(x_, st[name + ".kv.weight"], st[name + ".kv.bias"]).reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    attn = attn.softmax(dim=-1)
    attn = _m(masks, name + ".attn_drop", attn)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = F.linear(x, st[name + ".proj.weight"], st[name + ".proj.bias"])
    return _m(masks, name + ".proj_drop", x)


def dwconv_tokens(x, w, b, H, W):
    """DWConv.forward (ChangeFormer.py:517-523): tokens [B, N, C] -> NCHW -> depth-wise 3x3 (padding 1, bias) -> tokens.
    PINNED by tests/golden/g21_cf_decoder.npz (dw/*: the reference's own class, compiled from its file by ast)."""
    B, N, Ch = x.shape
    x = x.transpose(1, 2).reshape(B, Ch, H, W)
    x = F.conv2d(x, w, b, padding=1, groups=Ch)
    return x.flatten(2).transpose(1, 2)


def mix_ffn(st, name, x, H, W, masks):
    """Mlp.forward + DWConv.forward (ChangeFormer.py:287-295, 517-523)."""
    x = F.linear(x, st[name + ".fc1.weight"], st[name + ".fc1.bias"])
    x = dwconv_tokens(x, st[name + ".dwconv.dwconv.weight"], st[name + ".dwconv.dwconv.bias"], H, W)
    x = F.gelu(x)
    x = _m(masks, name + ".drop1", x)
    x = F.linear(x, st[name + ".fc2.weight"], st[name + ".fc2.bias"])
    return _m(masks, name + ".drop2", x)


def block(st, name, x, H, W, heads, sr, masks, tap=None):
    """Block.forward (ChangeFormer.py:505-509); drop_path is ONE module called twice: two independent draws."""
    C = x.shape[2]
    xn = F.layer_norm(x, (C,), st[name + ".norm1.weight"], st[name + ".norm1.bias"], 1e-6)
    y = attention(st, name + ".attn", xn, H, W, heads, sr, masks)
    x = x + _m(masks, name + ".drop_path1", y)
    if tap:
        _tap(tap + ".xn", xn, (H, W)); _tap(tap + ".pr", y, (H, W)); _tap(tap + ".x1", x, (H, W))
    y = mix_ffn(st, name + ".mlp", F.layer_norm(x, (C,), st[name + ".norm2.weight"], st[name + ".norm2.bias"], 1e-6), H, W, masks)
    x = x + _m(masks, name + ".drop_path2", y)
    if tap:
        _tap(tap + ".f2", y, (H, W)); _tap(tap + ".x2", x, (H, W))
    return x


def encoder(cfg: CFConfig, st, x, masks):
    """EncoderTransformer_v3.forward_features (ChangeFormer.py:1443-1470) -> 4 NCHW feature maps."""
    outs = []
    B = x.shape[0]
    for s in range(4):
        k, stride = (cfg.patch1, 4) if s == 0 else (cfg.patch, 2)
        t, H, W = patch_embed(st, f"Tenc_x2.patch_embed{s + 1}", x, k, stride)
        _tap(f"stage{s + 1}.tok", t, (H, W))
        for i in range(cfg.depths[s]):
            t = block(st, f"Tenc_x2.block{s + 1}.{i}", t, H, W, cfg.num_heads[s], cfg.sr_ratios[s], masks, f"stage{s + 1}.block{i}")
        C = t.shape[2]
        t = F.layer_norm(t, (C,), st[f"Tenc_x2.norm{s + 1}.weight"], st[f"Tenc_x2.norm{s + 1}.bias"], 1e-6)
        _tap(f"stage{s + 1}.out", t, (H, W))
        x = t.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


def _conv_diff(st, name, x, training, masks, tap=None):
    """conv_diff (ChangeFormer.py:1138-1148): Conv - PReLU - BN - Dropout(0.6) - Conv - PReLU - BN - Dropout(0.6)."""
    x = F.conv2d(x, st[name + ".0.weight"], st[name + ".0.bias"], padding=1)
    if tap:
        _tap(tap + ".ya", x)
    x = F.prelu(x, st[name + ".1.weight"])
    x = _m(masks, name + ".3", _bn(st, name + ".2", x, training))
    if tap:
        _tap(tap + ".aa", x)
    x = F.conv2d(x, st[name + ".4.weight"], st[name + ".4.bias"], padding=1)
    if tap:
        _tap(tap + ".yb", x)
    x = F.prelu(x, st[name + ".5.weight"])
    return _m(masks, name + ".7", _bn(st, name + ".6", x, training))


def _make_pred(st, name, x, training):
    """make_prediction (ChangeFormer.py:1151-1157): Conv - ReLU - BN - Conv."""
    x = F.relu(F.conv2d(x, st[name + ".0.weight"], st[name + ".0.bias"], padding=1))
    x = _bn(st, name + ".2", x, training)
    return F.conv2d(x, st[name + ".3.weight"], st[name + ".3.bias"], padding=1)


def decoder(cfg: CFConfig, st, f1, f2, training, masks):
    """DecoderTransformer_v3.forward (ChangeFormer.py:1563-1631) -> [p_c4, p_c3, p_c2, p_c1, cp]."""
    d = "TDec_x2."
    size1 = f1[0].shape[2:]
    outs, prev, ups = [], None, []
    for s in (4, 3, 2, 1):
        a, b = f1[s - 1], f2[s - 1]
        n, _, h, w = a.shape

        def mlp(x):      # MLP.forward (:685-688) + the permute / reshape of :1580
            y = F.linear(x.flatten(2).transpose(1, 2), st[d + f"linear_c{s}.proj.weight"], st[d + f"linear_c{s}.proj.bias"])
            return y.permute(0, 2, 1).reshape(n, -1, h, w)

        cat = torch.cat((mlp(a), mlp(b)), dim=1)
        _tap(f"dec.c{s}.cat", cat)
        c = _conv_diff(st, d + f"diff_c{s}", cat, training, masks, f"dec.c{s}")
        if prev is not None:
            c = c + F.interpolate(prev, scale_factor=2, mode="bilinear")
        _tap(f"dec.c{s}.c", c)
        outs.append(_make_pred(st, d + f"make_pred_c{s}", c, training))
        ups.append(c if s == 1 else F.interpolate(c, size=size1, mode="bilinear", align_corners=False))
        prev = c
    _tap("dec.fcat", torch.cat(ups, dim=1))
    x = F.conv2d(torch.cat(ups, dim=1), st[d + "linear_fuse.0.weight"], st[d + "linear_fuse.0.bias"])
    _tap("dec.fy", x)
    x = _bn(st, d + "linear_fuse.1", x, training)
    _tap("dec.fa", x)
    x = upsample_conv(x, st[d + "convd2x.conv2d.weight"], st[d + "convd2x.conv2d.bias"])
    _tap("dec.up2", x)
    x = residual_block(x, st[d + "dense_2x.0.conv1.conv2d.weight"], st[d + "dense_2x.0.conv1.conv2d.bias"],
                       st[d + "dense_2x.0.conv2.conv2d.weight"], st[d + "dense_2x.0.conv2.conv2d.bias"])
    _tap("dec.res2", x)
    x = upsample_conv(x, st[d + "convd1x.conv2d.weight"], st[d + "convd1x.conv2d.bias"])
    _tap("dec.up1", x)
    x = residual_block(x, st[d + "dense_1x.0.conv1.conv2d.weight"], st[d + "dense_1x.0.conv1.conv2d.bias"],
                       st[d + "dense_1x.0.conv2.conv2d.weight"], st[d + "dense_1x.0.conv2.conv2d.bias"])
    _tap("dec.res1", x)
    outs.append(conv_layer(x, st[d + "change_probability.conv2d.weight"], st[d + "change_probability.conv2d.bias"], 1))
    return outs


def forward(cfg: CFConfig, st, x1, x2, training: bool, masks: Optional[Dict[str, torch.Tensor]] = None) -> List[torch.Tensor]:
    """ChangeFormerV6.forward (ChangeFormer.py:1693-1701).  `masks` (training only): site name -> mask for BOTH dates stacked
    on dim 0 (date 1 rows, then date 2 rows) for encoder sites, [B, C, h, w] for the decoder's; missing names = no dropout."""
    B = x1.shape[0]
    m1 = m2 = None
    if training and masks is not None:
        m1 = {k: (v[:B] if k.startswith("Tenc_x2.") else v) for k, v in masks.items()}
        m2 = {k: (v[B:] if k.startswith("Tenc_x2.") else v) for k, v in masks.items()}
    f1 = encoder(cfg, st, x1, m1)
    f2 = encoder(cfg, st, x2, m2)
    return decoder(cfg, st, f1, f2, training, masks if training else None)


# ------------------------------------------------------------------------------------------------ the engine's mask hash
def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def site_seed(seed: int, site: int) -> int:
    """splitmix64(seed + (site + 1) * golden) >> 32: the 32-bit seed of one dropout site (stcd_cf_site_seed)."""
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (site + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return int(z >> 32)


def hash_keep(n: int, seed32: int, p: float) -> np.ndarray:
    """keep[i] for i < n: (fmix32(i * 0x9E3779B1 + seed32) >> 8) >= floor(p * 2^24) -- the engine's per-element rule."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64).astype(np.uint32)
        h = _fmix32((idx * np.uint32(0x9E3779B1) + np.uint32(seed32)).astype(np.uint32))
    thr = np.uint32(int(np.float32(p) * np.float32(16777216.0)))
    return (h >> np.uint32(8)) >= thr


def site_list(cfg: CFConfig, B: int, H: int, W: int):
    """[(name, engine-layout shape, p)] in the engine's site order (stcd_cf_site_get): encoder sites hold both dates
    (2B images, date 1 first); element order is the engine's NHWC / [n, heads, N, Nkv] linear index."""
    sites = []
    dpr = cfg.dpr()
    h, w = H, W
    j = 0
    for s in range(4):
        stride = 4 if s == 0 else 2
        k = cfg.patch1 if s == 0 else cfg.patch
        h = (h + 2 * (k // 2) - k) // stride + 1
        w = (w + 2 * (k // 2) - k) // stride + 1
        C, sr, heads = cfg.embed_dims[s], cfg.sr_ratios[s], cfg.num_heads[s]
        N = h * w
        Nkv = (h // sr) * (w // sr) if sr > 1 else N
        for i in range(cfg.depths[s]):
            b = f"Tenc_x2.block{s + 1}.{i}"
            sites.append((b + ".attn.attn_drop", (2 * B, heads, N, Nkv), cfg.attn_drop))
            sites.append((b + ".attn.proj_drop", (2 * B, N, C), cfg.drop_rate))
            sites.append((b + ".drop_path1", (2 * B,), dpr[j]))
            sites.append((b + ".mlp.drop1", (2 * B, N, cfg.mlp_ratio * C), cfg.drop_rate))
            sites.append((b + ".mlp.drop2", (2 * B, N, C), cfg.drop_rate))
            sites.append((b + ".drop_path2", (2 * B,), dpr[j]))
            j += 1
    hs, ws = [], []
    h, w = H, W
    for s in range(4):
        stride = 4 if s == 0 else 2
        k = cfg.patch1 if s == 0 else cfg.patch
        h = (h + 2 * (k // 2) - k) // stride + 1
        w = (w + 2 * (k // 2) - k) // stride + 1
        hs.append(h)
        ws.append(w)
    for s in (4, 3, 2, 1):
        for sub in ("3", "7"):
            sites.append((f"TDec_x2.diff_c{s}.{sub}", (B, hs[s - 1], ws[s - 1], cfg.embedding_dim), cfg.diff_drop))
    return sites


def engine_masks(cfg: CFConfig, B: int, H: int, W: int, seed: int, dtype=torch.float32, sites=None) -> Dict[str, torch.Tensor]:
    """The masks the HIP engine draws for `seed` (counter hash per site), in this file's layouts.  `sites`: the engine's own table
    (Engine.cf_sites(): the DropPath probabilities as the engine's fp32 arithmetic produced them); default: site_list(cfg)."""
    out = {}
    for site, (name, shape, p) in enumerate(sites if sites is not None else site_list(cfg, B, H, W)):
        if p <= 0.0:
            continue
        n = int(np.prod(shape))
        keep = hash_keep(n, site_seed(seed, site), p).reshape(shape)
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        m = torch.from_numpy(keep.astype(np.float32) * scale).to(dtype)
        if len(shape) == 1:                       # DropPath: per sample, broadcast over [N, C]
            m = m.reshape(-1, 1, 1)
        elif name.startswith("TDec_x2."):         # engine NHWC -> NCHW
            m = m.permute(0, 3, 1, 2).contiguous()
        out[name] = m
    return out


def random_masks(cfg: CFConfig, B: int, H: int, W: int, seed: int) -> Dict[str, torch.Tensor]:
    """Independent Bernoulli masks in the same layouts (oracle-only experiments)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, p in site_list(cfg, B, H, W):
        if p <= 0.0:
            continue
        m = (torch.rand(shape, generator=g) >= p).float() / (1.0 - p)
        if len(shape) == 1:
            m = m.reshape(-1, 1, 1)
        elif name.startswith("TDec_x2."):
            m = m.permute(0, 3, 1, 2).contiguous()
        out[name] = m
    return out
