"""Drop-in ``ChangeFormerV6`` over the HIP engine (BASELINE.json configs[4]).

Same constructor, ``forward(x1, x2) -> [p_c4, p_c3, p_c2, p_c1, cp]`` contract, parameter / buffer names, shapes and registration
order as the reference class (``state_dict`` files interchange both ways; ``define_G("ChangeFormerV6")`` builds this one):

    ChangeFormerV6(input_nc=3, output_nc=2, decoder_softmax=False, embed_dim=256)   /root/reference/models/ChangeFormer.py:1669-1701
    EncoderTransformer_v3 / OverlapPatchEmbed / Block / Attention / Mlp / DWConv      ChangeFormer.py:1342-1473, 195-236, 472-523, 260-358
    DecoderTransformer_v3 / MLP / conv_diff / make_prediction                         ChangeFormer.py:1475-1631, 677-688, 1138-1157
    ConvLayer / UpsampleConvLayer / ResidualBlock                                     /root/reference/models/ChangeFormerBaseNetworks.py:85-120

The sub-modules are ordinary torch layer objects used purely as parameter holders (never called): all arithmetic of forward and
backward runs in libstcd_hip.so, there is no eager fallback and CPU tensors are rejected.  What timm supplies in the reference is
restated from its published definitions: ``DropPath`` (a per-sample Bernoulli(keep) / keep mask -- here a counter hash inside the
engine), ``trunc_normal_`` (``torch.nn.init.trunc_normal_``).

Limits (stated, and raised loudly): only the LAST output (cp) carries a gradient -- the reference's default loss uses
``G_pred[-1]`` (models/trainer.py:311, ``multi_scale_train == "False"``); a gradient into the four auxiliary maps raises.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from ._lib import StcdError
from .modules import HipChangeDetector

V6 = dict(embed_dims=(64, 128, 320, 512), depths=(3, 3, 4, 3), num_heads=(1, 2, 4, 8), sr_ratios=(8, 4, 2, 1), mlp_ratio=4,
          patch1=7, patch=7, drop_rate=0.1, attn_drop=0.1, drop_path_rate=0.1)
# The encoder BASELINE.json configs[4] names ("ChangeFormer (MiT-B0 encoder)"; SURVEY R10: not what the in-tree V6 builds): the mit_b0
# entry of the reference's vendored encoder zoo, /root/reference/segmentation_models_pytorch/encoders/mix_transformer.py:497-511 --
# widths [32, 64, 160, 256], depths [2, 2, 2, 2], heads [1, 2, 5, 8] (head dimension 32 everywhere), patch 7 / 3 / 3 / 3, drop 0.0,
# drop_path 0.1.  ChangeFormerV6(config=MIT_B0) runs it under the same decoder.
MIT_B0 = dict(embed_dims=(32, 64, 160, 256), depths=(2, 2, 2, 2), num_heads=(1, 2, 5, 8), sr_ratios=(8, 4, 2, 1), mlp_ratio=4,
              patch1=7, patch=3, drop_rate=0.0, attn_drop=0.0, drop_path_rate=0.1)


class DropPath(nn.Module):
    """Holder with timm's attribute name (``drop_prob``); the engine draws the per-sample mask."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob


class _DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, drop):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = _DWConv(hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, attn_drop, proj_drop, sr_ratio):
        super().__init__()
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.kv = nn.Linear(dim, dim * 2, bias=True)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, drop, attn_drop, drop_path, sr_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads, attn_drop, drop, sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), drop)


class _PatchEmbed(nn.Module):
    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)


def _encoder_init(m):
    """EncoderTransformer_v3._init_weights (ChangeFormer.py:1411-1424)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


class _Encoder(nn.Module):
    def __init__(self, in_chans, cfg):
        super().__init__()
        E, depths = cfg["embed_dims"], cfg["depths"]
        self.depths, self.embed_dims = list(depths), list(E)
        cin = in_chans
        for s in range(4):
            k, stride = (cfg["patch1"], 4) if s == 0 else (cfg["patch"], 2)
            setattr(self, f"patch_embed{s + 1}", _PatchEmbed(k, stride, cin, E[s]))
            cin = E[s]
        dpr = [x.item() for x in torch.linspace(0, cfg["drop_path_rate"], sum(depths))]
        cur = 0
        for s in range(4):
            setattr(self, f"block{s + 1}", nn.ModuleList([
                _Block(E[s], cfg["num_heads"][s], cfg["mlp_ratio"], cfg["drop_rate"], cfg["attn_drop"], dpr[cur + i], cfg["sr_ratios"][s])
                for i in range(depths[s])]))
            setattr(self, f"norm{s + 1}", nn.LayerNorm(E[s], eps=1e-6))
            cur += depths[s]
        self.apply(_encoder_init)


class _MLP(nn.Module):
    def __init__(self, input_dim, embed_dim):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)


class _ConvLayer(nn.Module):
    def __init__(self, cin, cout, k, stride, padding):
        super().__init__()
        self.conv2d = nn.Conv2d(cin, cout, k, stride, padding)


class _UpsampleConvLayer(nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.conv2d = nn.ConvTranspose2d(cin, cout, k, stride=stride, padding=1)


class _ResidualBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = _ConvLayer(c, c, 3, 1, 1)
        self.conv2 = _ConvLayer(c, c, 3, 1, 1)
        self.relu = nn.ReLU()


def _conv_diff(cin, cout, p=0.6):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.PReLU(), nn.BatchNorm2d(cout), nn.Dropout(p=p),
                         nn.Conv2d(cout, cout, kernel_size=3, padding=1), nn.PReLU(), nn.BatchNorm2d(cout), nn.Dropout(p=p))


def _make_prediction(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.ReLU(), nn.BatchNorm2d(cout),
                         nn.Conv2d(cout, cout, kernel_size=3, padding=1))


class _Decoder(nn.Module):
    def __init__(self, in_channels, embedding_dim, output_nc, decoder_softmax):
        super().__init__()
        D = embedding_dim
        self.embedding_dim, self.output_nc, self.output_softmax = D, output_nc, decoder_softmax
        for s in (4, 3, 2, 1):
            setattr(self, f"linear_c{s}", _MLP(in_channels[s - 1], D))
        for s in (4, 3, 2, 1):
            setattr(self, f"diff_c{s}", _conv_diff(2 * D, D))
        for s in (4, 3, 2, 1):
            setattr(self, f"make_pred_c{s}", _make_prediction(D, output_nc))
        self.linear_fuse = nn.Sequential(nn.Conv2d(D * len(in_channels), D, kernel_size=1), nn.BatchNorm2d(D))
        self.convd2x = _UpsampleConvLayer(D, D, 4, 2)
        self.dense_2x = nn.Sequential(_ResidualBlock(D))
        self.convd1x = _UpsampleConvLayer(D, D, 4, 2)
        self.dense_1x = nn.Sequential(_ResidualBlock(D))
        self.change_probability = _ConvLayer(D, output_nc, 3, 1, 1)
        self.active = nn.Sigmoid()


class ChangeFormerV6(HipChangeDetector):
    """ChangeFormerV6(input_nc=3, output_nc=2, decoder_softmax=False, embed_dim=256).forward(x1, x2) -> list of five maps.

    ``config`` (keyword only, not in the reference) overrides the V6 widths / depths / heads / sr ratios / drop rates with another
    member of the same class family (e.g. a small one for tests); the default is the reference's V6."""

    ARCH = "changeformer"
    RETURNS_LIST = True
    OUT_MAPS = 0           # five maps of different sizes in one flat buffer

    def __init__(self, input_nc=3, output_nc=2, decoder_softmax=False, embed_dim=256, dtype: Optional[str] = None, *,
                 config: Optional[dict] = None):
        cfg = dict(V6)
        cfg.update(config or {})
        cfg["embedding_dim"] = embed_dim
        cfg.setdefault("diff_drop", 0.6)
        super().__init__(input_nc, output_nc, dtype, cfg)
        self._cfg = cfg
        self.embed_dims, self.depths, self.embedding_dim = list(cfg["embed_dims"]), list(cfg["depths"]), embed_dim
        self.drop_rate, self.attn_drop, self.drop_path_rate = cfg["drop_rate"], cfg["attn_drop"], cfg["drop_path_rate"]
        self.Tenc_x2 = _Encoder(input_nc, cfg)
        self.TDec_x2 = _Decoder(self.embed_dims, embed_dim, output_nc, decoder_softmax)
        for s in (4, 3, 2, 1):
            for idx in (3, 7):
                getattr(self.TDec_x2, f"diff_c{s}")[idx].p = cfg["diff_drop"]
        self._check_layout()

    def __deepcopy__(self, memo):
        eng = self._engine
        new = type(self)(eng.in_ch, eng.label_ch, self.TDec_x2.output_softmax, self.embedding_dim, eng.dtype, config=self._cfg)
        new.load_state_dict({k: v.detach().clone() for k, v in self.state_dict().items()})
        new.train(self.training)
        if self._flat_params is not None:
            new.to(self._flat_params.device)
        return new

    def set_dropout_p(self, p: float):
        raise StcdError("ChangeFormer has several drop rates: use set_drop_rates(drop_rate, attn_drop, diff_drop)")

    def set_drop_rates(self, drop_rate: float, attn_drop: float, diff_drop: float = 0.6):
        """Element-wise dropout rates (Mlp / proj dropout, attention dropout, conv_diff dropout); 0 disables a family."""
        self._engine.cf_set_drop_rates(drop_rate, attn_drop, diff_drop)
        for m in self.Tenc_x2.modules():
            if isinstance(m, _Attention):
                m.attn_drop.p, m.proj_drop.p = attn_drop, drop_rate
            elif isinstance(m, _Mlp):
                m.drop.p = drop_rate
        for s in (4, 3, 2, 1):
            for idx in (3, 7):
                getattr(self.TDec_x2, f"diff_c{s}")[idx].p = diff_drop

    def set_dropout_masks(self, masks):
        raise StcdError("ChangeFormer masks come from the engine's counter hash of the step seed (set_seed); "
                        "oracle.changeformer_ref.engine_masks reproduces them")

    def set_seed(self, seed: int):
        """The NEXT training forward draws its masks from exactly this seed (parity tests)."""
        self._forced_seed = int(seed)

    def last_seed(self) -> int:
        return self._last_seed

    def _run_forward(self, x1, x2, training: bool):
        forced = getattr(self, "_forced_seed", None)
        if forced is not None and training:
            eng = self._engine
            logits = torch.empty(eng.output_floats(), dtype=torch.float32, device=x1.device)
            self._steps += 1
            eng.forward(x1, x2, self._flat_params, self._flat_bn, logits, training, None, seed=forced)
            self._nbt.add_(self._nbt_inc)
            self._last_seed, self._forced_seed = forced, None
            return logits
        self._last_seed = (self._seed * 1000003 + self._steps + 1) & (2 ** 64 - 1)
        return super()._run_forward(x1, x2, training)

    # five maps of different sizes share one flat buffer
    def _split_outputs(self, flat) -> List[torch.Tensor]:
        B = self._engine.shape[0]
        L = self._engine.label_ch
        return [flat[off:off + B * L * h * w].view(B, L, h, w) for off, h, w in self._engine.cf_outputs()]

    def set_multi_scale_train(self, on: bool = True):
        """The reference's `multi_scale_train == "True"` (models/trainer.py:300-309: the loss is a weighted sum over all five
        predictions): plan the backward of the four auxiliary heads as well.  Off by default, as in the reference (the heads'
        weight-gradient jobs and scratch maps are then not part of the step at all)."""
        self._engine.cf_set_aux_backward(bool(on))
        self._aux_bwd = bool(on)

    def _merge_grads(self, flat, grads):
        """Gradients of the five outputs in the engine's output layout."""
        aux = any(g is not None for g in grads[:4])
        if aux and not getattr(self, "_aux_bwd", False):
            raise StcdError("ChangeFormer: a gradient arrived for the auxiliary maps p_c4..p_c1 (a multi-scale loss, trainer.py:300-309) but "
                            "the engine was planned for the default cp-only loss: call model.set_multi_scale_train(True) before the step")
        g = torch.zeros_like(flat)
        outs = self._engine.cf_outputs()
        for i in range(5):
            if grads[i] is not None:
                off = outs[i][0]
                g[off:off + grads[i].numel()].copy_(grads[i].reshape(-1))
        return g

    def _wrap_output(self, out, B):
        outs = list(out)
        if self.TDec_x2.output_softmax:
            outs = [torch.sigmoid(o) for o in outs]
        return outs
