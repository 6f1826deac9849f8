"""Drop-in ``torch.nn.Module`` boundary over the HIP engine.

Same constructor signatures, ``forward(x1, x2)`` contract, parameter / buffer names and shapes as the
reference classes, so ``state_dict`` files interchange both ways and ``define_G`` / ``init_weights`` /
optimizers / ``nn.DataParallel([0])`` work unchanged:

    SiamUnet_diff(input_nbr, label_nbr)   /root/reference/models/SiamUnet_diff.py:13,94   -> tensor
    SiamUnet_conc(input_nbr, label_nbr)   /root/reference/models/SiamUnet_conc.py:13,94   -> tensor
    SiamUnet_sub(input_nbr, label_nbr)    /root/reference/models/SiamUnet_sub.py:13,94    -> [tensor]

The sub-modules (``conv11``, ``bn11``, ``do11`` ...) are ordinary torch layer objects used purely as
parameter holders -- they are never called.  All arithmetic of forward and backward runs in
libstcd_hip.so; there is no eager/PyTorch fallback, and CPU tensors are rejected.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from ._lib import StcdError
from .engine import Engine

_ENC = ((("11", None, 16), ("12", 16, 16)),
        (("21", 16, 32), ("22", 32, 32)),
        (("31", 32, 64), ("32", 64, 64), ("33", 64, 64)),
        (("41", 64, 128), ("42", 128, 128), ("43", 128, 128)))
_DEC = (("upconv4", 128, (("43d", 128), ("42d", 128), ("41d", 64))),
        ("upconv3", 64, (("33d", 64), ("32d", 64), ("31d", 32))),
        ("upconv2", 32, (("22d", 32), ("21d", 16))),
        ("upconv1", 16, (("12d", 16), ("11d", None))))


def default_dtype() -> str:
    return os.environ.get("STCD_DTYPE", "bf16")


class _EngineFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are single engine calls."""

    @staticmethod
    def forward(ctx, model, anchor, x1, x2):
        logits = model._run_forward(x1, x2, True)
        ctx.model = model
        ctx.ticket = model._engine.ticket
        n = model.OUT_MAPS
        if n == 0:      # maps of different sizes in one flat buffer (ChangeFormer): the model splits / merges them
            ctx.set_materialize_grads(False)
            ctx.flat = logits
            return tuple(model._split_outputs(logits))
        if n == 1:
            return logits
        # several maps per pair (SegCD): one autograd output each, so an unused map costs nothing in the backward (slicing ONE
        # output tensor outside made autograd zero-fill and copy through ~45 small kernels per step)
        ctx.set_materialize_grads(False)
        ctx.out_shape = logits.shape
        b = logits.shape[0] // n
        return tuple(logits[k * b:(k + 1) * b] for k in range(n))

    @staticmethod
    def backward(ctx, *grads):
        model = ctx.model
        if ctx.ticket != model._engine.ticket:
            raise StcdError("backward() for a forward pass whose saved activations were overwritten by a later "
                            "forward of the same module (the engine keeps one step of activations)")
        if model.OUT_MAPS == 0:
            g = model._merge_grads(ctx.flat, grads)
        elif len(grads) == 1:
            g = grads[0].contiguous()
        else:
            live = [gk for gk in grads if gk is not None]
            g = torch.empty(ctx.out_shape, dtype=torch.float32, device=live[0].device)
            b = ctx.out_shape[0] // len(grads)
            for k, gk in enumerate(grads):
                if gk is None:
                    g[k * b:(k + 1) * b].zero_()
                else:
                    g[k * b:(k + 1) * b].copy_(gk)
        model._run_backward(g)
        return None, None, None, None


class _NoEvalGradFn(torch.autograd.Function):
    """Eval-mode outputs carry no saved activations: make a backward through them fail loudly instead of silently
    handing autograd a constant (the engine's backward is the training-mode BatchNorm backward only)."""

    @staticmethod
    def forward(ctx, out, anchor):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        raise StcdError("backward() through an eval-mode forward: the HIP engine differentiates training-mode forwards only "
                        "(call .train(), or wrap inference in torch.no_grad()); see INTEGRATION.md, limits")


def frozen_weights(model):
    """``model.frozen_weights()`` for an engine module (also behind ``nn.DataParallel``), a no-op context for anything else."""
    import contextlib
    m = getattr(model, "module", model)
    return m.frozen_weights() if isinstance(m, HipChangeDetector) else contextlib.nullcontext(model)


class HipChangeDetector(nn.Module):
    """Common machinery: flat parameter / gradient / BN buffers shared with the engine."""

    ARCH = None
    RETURNS_LIST = False
    OUT_MAPS = 1          # maps per pair in the engine's output buffer ([OUT_MAPS*B, label, H, W])

    def __init__(self, in_ch: int, label_ch: int, dtype: Optional[str] = None, cf_config: Optional[dict] = None):
        super().__init__()
        self._engine = Engine(self.ARCH, in_ch, label_ch, dtype or default_dtype(), cf_config)
        self._flat_params: Optional[torch.Tensor] = None
        self._flat_grads: Optional[torch.Tensor] = None
        self._flat_bn: Optional[torch.Tensor] = None
        self._anchor: Optional[torch.Tensor] = None
        self._grad_views: List[torch.Tensor] = []
        self._next_masks: Optional[torch.Tensor] = None
        self._seed = int(os.environ.get("STCD_DROPOUT_SEED", "1337"))
        self._steps = 0
        self._grad_stage_hook: Optional[Callable[[int, torch.Tensor], None]] = None

    @property
    def grad_stage_hook(self) -> Optional[Callable[[int, torch.Tensor], None]]:
        """hook(stage, flat_gradient_slice), called when that slice is final (stcd_amd.ddp.FlatGradReducer).  With a hook the
        backward runs as two staged calls on ONE stream, so the engine's side stream for the decoder's weight gradients (and the
        smaller grids planned for it) is switched off: the plan is rebuilt by the next forward."""
        return self._grad_stage_hook

    @grad_stage_hook.setter
    def grad_stage_hook(self, hook):
        changed = (hook is None) != (self._grad_stage_hook is None)
        self._grad_stage_hook = hook
        eng = getattr(self, "_engine", None)
        if changed and eng is not None and hasattr(eng, "set_wgrad_side"):
            eng.set_wgrad_side(hook is None)

    # ------------------------------------------------------------------ parameter plumbing
    def _check_layout(self):
        names = [n for n, _ in self.named_parameters()]
        want = [p.name for p in self._engine.params]
        if names != want:
            raise StcdError(f"module/engine parameter order mismatch: {names[:4]}... vs {want[:4]}...")
        for (n, p), info in zip(self.named_parameters(), self._engine.params):
            if tuple(p.shape) != info.shape:
                raise StcdError(f"shape mismatch for {n}: {tuple(p.shape)} vs {info.shape}")

    def _bn_modules(self):
        mods = dict(self.named_modules())
        return [(b, mods[b.name]) for b in self._engine.bns]

    def _param_slots(self):
        """[(owning module, key, Parameter)] in registration order: the per-step checks walk THIS list (dictionary look-ups) instead of
        nn.Module's generator traversal -- `self.parameters()` costs ~1.5 ms per call on ChangeFormer's 401 tensors, and a step made
        three such walks (host profile).  A re-assigned Parameter object or `.data` is still noticed."""
        slots = []
        seen = set()
        for mod in self.modules():
            for key, prm in mod._parameters.items():
                if prm is not None and id(prm) not in seen:
                    seen.add(id(prm))
                    slots.append((mod, key, prm))
        return slots

    def _live_params(self):
        """The module's parameters through the cached slots (None when the module tree changed: callers fall back to a full walk)."""
        slots = getattr(self, "_pslots", None)
        if slots is None:
            return None
        out = []
        for mod, key, prm in slots:
            if mod._parameters.get(key) is not prm:
                return None
            out.append(prm)
        return out

    def _views_ok(self, device) -> bool:
        fp = self._flat_params
        if fp is None or fp.device != device:
            return False
        params = self._live_params()
        if params is None or len(params) != len(self._engine.params):
            return False
        base = fp.data_ptr()
        for p, info in zip(params, self._engine.params):      # every parameter: a re-assigned p.data is noticed
            if p.data_ptr() != base + 4 * info.offset or p.dtype != torch.float32:
                return False
        b, m = self._bn_cached[-1]
        return m._buffers.get("running_mean") is not None and m._buffers["running_mean"].data_ptr() == self._flat_bn.data_ptr() + 4 * b.offset

    def _ensure_flat(self, device):
        if self._views_ok(device):
            return
        eng = self._engine
        flat = torch.zeros(eng.param_floats, dtype=torch.float32, device=device)
        for p, info in zip(self.parameters(), eng.params):
            v = flat[info.offset:info.offset + info.numel].view(info.shape)
            v.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = v
            p.grad = None
        self._flat_params = flat
        self._flat_grads = torch.zeros_like(flat)
        self._grad_views = [self._flat_grads[i.offset:i.offset + i.numel].view(i.shape) for i in eng.params]
        fbn = torch.zeros(eng.bn_floats, dtype=torch.float32, device=device)
        for b, m in self._bn_modules():
            for k, name in enumerate(("running_mean", "running_var")):
                v = fbn[b.offset + k * b.channels:b.offset + (k + 1) * b.channels]
                v.copy_(getattr(m, name).to(device=device, dtype=torch.float32))
                m._buffers[name] = v
        # num_batches_tracked: 0-dim views of one int64 vector, bumped by ONE add per training forward
        bnm = self._bn_modules()
        nbt = torch.stack([m.num_batches_tracked.to(device=device, dtype=torch.int64).reshape(()) for _, m in bnm])
        for i, (_, m) in enumerate(bnm):
            m._buffers["num_batches_tracked"] = nbt[i]
        self._flat_bn = fbn
        self._nbt = nbt
        self._nbt_inc = torch.tensor([b.calls_per_forward for b, _ in bnm], dtype=torch.int64, device=device)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._pslots = self._param_slots()
        self._bn_cached = bnm
        if [id(p) for _, _, p in self._pslots] != [id(p) for p in self.parameters()]:
            raise StcdError("internal: parameter slot order differs from nn.Module.parameters()")

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flat_params = None   # .to()/.cuda()/.float() re-created the tensors: re-flatten lazily
        self._pslots = None
        return out

    def _weights_changed(self):
        """Inside frozen_weights(): the parameters were rewritten after all (load_state_dict, an optimizer step): take a new tag so
        the next forward repacks instead of computing with stale filter images."""
        if getattr(self, "_freeze_depth", 0) > 0:
            self._freeze_count = getattr(self, "_freeze_count", 0) + 1
            self._engine.set_weights_tag(self._freeze_count)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._weights_changed()
        return out

    def __deepcopy__(self, memo):
        new = type(self)(self._engine.in_ch, self._engine.label_ch, self._engine.dtype)
        new.load_state_dict({k: v.detach().clone() for k, v in self.state_dict().items()})
        new.train(self.training)
        new._engine_dropout_p(getattr(self, "_drop_p", 0.2))
        if self._flat_params is not None:
            new.to(self._flat_params.device)
        return new

    # ------------------------------------------------------------------ inference loops
    def frozen_weights(self):
        """Context manager for inference loops: inside it the caller vouches that the parameters do not change, so the engine
        packs its filter images once instead of on every forward (they are re-packed on every call otherwise, because an optimizer
        may have rewritten them -- through ``.data`` or a fused kernel, which nothing here could notice)::

            model.eval()
            with torch.no_grad(), model.frozen_weights():
                for a, b in loader:
                    out = model(a, b)
        """
        import contextlib

        @contextlib.contextmanager
        def _ctx():
            # re-entrant: nested contexts (a user loop already inside frozen_weights() calling CDTrainer's validation, which wraps
            # itself again) share the outermost tag; the vouch ends when the OUTERMOST context exits
            depth = getattr(self, "_freeze_depth", 0)
            if depth == 0:
                self._freeze_count = getattr(self, "_freeze_count", 0) + 1
                self._engine.set_weights_tag(self._freeze_count)
            self._freeze_depth = depth + 1
            try:
                yield self
            finally:
                self._freeze_depth -= 1
                if self._freeze_depth == 0:
                    self._engine.set_weights_tag(0)
        return _ctx()

    # ------------------------------------------------------------------ knobs
    def _engine_dropout_p(self, p: float):
        self._drop_p = p
        self._engine.set_dropout_p(p)

    def set_dropout_p(self, p: float):
        """Dropout2d probability (reference: 0.2); also reflected in the holder modules."""
        self._engine_dropout_p(p)
        for m in self.modules():
            if isinstance(m, nn.Dropout2d):
                m.p = p

    def set_dropout_masks(self, masks: Optional[Dict[str, torch.Tensor]]):
        """Use these [rows, C] masks ({0, 1/(1-p)}) verbatim for the NEXT training forward (parity tests)."""
        self._pending_masks = masks

    # ------------------------------------------------------------------ execution
    def _run_forward(self, x1, x2, training: bool):
        eng = self._engine
        B, _, H, W = x1.shape
        if self.OUT_MAPS == 0:
            logits = torch.empty(eng.output_floats(), dtype=torch.float32, device=x1.device)
        else:
            logits = torch.empty((self.OUT_MAPS * B, eng.label_ch, H, W), dtype=torch.float32, device=x1.device)
        masks = None
        pend = getattr(self, "_pending_masks", None)
        if training and pend is not None:
            masks = eng.pack_masks(pend, x1.device)
            self._pending_masks = None
        self._steps += 1
        eng.forward(x1, x2, self._flat_params, self._flat_bn, logits, training, masks,
                    seed=self._seed * 1000003 + self._steps)
        if training:
            self._nbt.add_(self._nbt_inc)
        return logits

    def _run_backward(self, grad_logits):
        eng = self._engine
        params = self._live_params() or list(self.parameters())
        accumulate = any(p.grad is not None for p in params)
        target = torch.empty_like(self._flat_grads) if accumulate else self._flat_grads
        hook = self.grad_stage_hook
        if hook is None:
            eng.backward(grad_logits, self._flat_params, target, -1)
        else:
            eng.backward(grad_logits, self._flat_params, target, 0)
            hook(0, target[eng.stage0_range[0]:eng.stage0_range[1]])
            eng.backward(grad_logits, self._flat_params, target, 1)
            hook(1, target[eng.stage1_range[0]:eng.stage1_range[1]])
        if accumulate:
            for p, info in zip(params, eng.params):
                g = target[info.offset:info.offset + info.numel].view(info.shape)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.add_(g)
        else:
            for p, v in zip(params, self._grad_views):
                p.grad = v

    def forward(self, x1, x2):
        if not (x1.is_cuda and x2.is_cuda):
            raise StcdError("the HIP engine needs inputs on a GPU (cuda/HIP device); there is no CPU fallback")
        if x1.shape != x2.shape or x1.dim() != 4 or x1.shape[1] != self._engine.in_ch:
            raise StcdError(f"expected two [B,{self._engine.in_ch},H,W] tensors, got {tuple(x1.shape)} and {tuple(x2.shape)}")
        dev = x1.device
        self._ensure_flat(dev)
        x1 = x1.detach().contiguous().float()
        x2 = x2.detach().contiguous().float()
        B, _, H, W = x1.shape
        with torch.cuda.device(dev):
            self._engine.configure(B, H, W, dev)
            if self.training and torch.is_grad_enabled():
                out = _EngineFn.apply(self, self._anchor, x1, x2)
            else:
                out = self._run_forward(x1, x2, self.training)
                guard = torch.is_grad_enabled() and any(p.requires_grad for p in (self._live_params() or self.parameters()))
                if self.OUT_MAPS == 0:
                    # ChangeFormer: five maps in one flat buffer; the guard sits on the flat buffer so every map carries it
                    if guard:
                        out = _NoEvalGradFn.apply(out, self._anchor)
                    out = tuple(self._split_outputs(out))
                elif torch.is_grad_enabled() and any(p.requires_grad for p in (self._live_params() or self.parameters())):
                    out = _NoEvalGradFn.apply(out, self._anchor)
        return self._wrap_output(out, B)

    def _wrap_output(self, out, B):
        return [out] if self.RETURNS_LIST else out


class _CrossConcHolder(nn.Module):
    """Parameter holder of the reference's `cross_conc` block (SiamUnet_crossconc.py:11-22); the engine runs it."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels = in_channels
        self.diff = nn.Sequential(nn.Conv2d(in_channels, in_channels // 2, kernel_size=3, padding=1, stride=1, groups=in_channels // 2),
                                  nn.BatchNorm2d(in_channels // 2), nn.ReLU())
        self.conv_res = nn.Sequential(nn.Conv2d(in_channels // 2, out_channels, kernel_size=3, padding=1, stride=1), nn.BatchNorm2d(out_channels))
        self.act = nn.ReLU()


class _FCSiam(HipChangeDetector):
    """Layer holders in the reference's registration order (SiamUnet_diff.py:18-90)."""

    def __init__(self, input_nbr, label_nbr, dtype: Optional[str] = None):
        super().__init__(input_nbr, label_nbr, dtype)
        self.input_nbr = input_nbr
        for stage in _ENC:
            for sfx, ci, co in stage:
                ci = (2 * input_nbr if self.ARCH == "fcef" else input_nbr) if ci is None else ci
                setattr(self, f"conv{sfx}", nn.Conv2d(ci, co, kernel_size=3, padding=1))
                setattr(self, f"bn{sfx}", nn.BatchNorm2d(co))
                setattr(self, f"do{sfx}", nn.Dropout2d(p=0.2))
        for up, c, convs in _DEC:
            setattr(self, up, nn.ConvTranspose2d(c, c, kernel_size=3, padding=1, stride=2, output_padding=1))
            ci = c + (2 * c if self.ARCH == "conc" else c)
            for sfx, co in convs:
                co_ = label_nbr if co is None else co
                setattr(self, f"conv{sfx}", nn.ConvTranspose2d(ci, co_, kernel_size=3, padding=1))
                if co is not None:
                    setattr(self, f"bn{sfx}", nn.BatchNorm2d(co))
                    setattr(self, f"do{sfx}", nn.Dropout2d(p=0.2))
                ci = co_
        self.sm = nn.LogSoftmax(dim=1)   # present (unused) in the reference too: SiamUnet_diff.py:92
        if self.ARCH == "xconc":         # SiamUnet_crossconc.py:119-122
            for l, c in enumerate((16, 32, 64, 128), 1):
                setattr(self, f"cross_conc{l}", _CrossConcHolder(2 * c, c))
        self._check_layout()


class SiamUnet_diff(_FCSiam):
    """FC-Siam-diff: skips = |f1 - f2| (SiamUnet_diff.py:150)."""
    ARCH = "diff"


class SiamUnet_conc(_FCSiam):
    """FC-Siam-conc: skips = cat(f1, f2) (SiamUnet_conc.py:149)."""
    ARCH = "conc"


class SiamUnet_sub(_FCSiam):
    """Signed skips f2 - f1; returns a one-element list like the reference (SiamUnet_sub.py:150,177-180)."""
    ARCH = "sub"
    RETURNS_LIST = True


class SiamUnet_cross_conc(_FCSiam):
    """FC-Siam with a `cross_conc` block on every skip (SiamUnet_crossconc.py:35-212): cat(upsampled, cross_conc_l(f1, f2)); returns a
    one-element list like the reference (:206-208)."""
    ARCH = "xconc"
    RETURNS_LIST = True


class Unet(_FCSiam):
    """FC-EF (models/Unet.py:10-154): ONE encoder stream over cat(x1, x2) (conv11 takes 2 * input_nbr channels), skips = the
    stream's own activations (Unet.py:128,136,144,151), SiamUnet_diff's decoder; returns the logits tensor."""
    ARCH = "fcef"


class _NestedBlockHolder(nn.Module):
    """Parameter holder with the reference's names: conv_block_nested (SNUNet.py:8-16)."""

    def __init__(self, in_ch, mid_ch, out_ch):
        super().__init__()
        self.activation = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_ch, mid_ch, kernel_size=3, padding=1, bias=True)
        self.bn1 = nn.BatchNorm2d(mid_ch)
        self.conv2 = nn.Conv2d(mid_ch, out_ch, kernel_size=3, padding=1, bias=True)
        self.bn2 = nn.BatchNorm2d(out_ch)


class _UpHolder(nn.Module):
    """up (SNUNet.py:29-38): ConvTranspose2d(C, C, 2, stride=2)."""

    def __init__(self, in_ch):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_ch, in_ch, 2, stride=2)


class _ChannelAttentionHolder(nn.Module):
    """ChannelAttention (SNUNet.py:46-54): two bias-free 1x1 convs."""

    def __init__(self, in_channels, ratio=16):
        super().__init__()
        self.fc1 = nn.Conv2d(in_channels, in_channels // ratio, 1, bias=False)
        self.fc2 = nn.Conv2d(in_channels // ratio, in_channels, 1, bias=False)


class SNUNet_ECAM(HipChangeDetector):
    """SNUNet-CD with ECAM (SNUNet.py:63-152): SNUNet_ECAM(in_ch=3, out_ch=1).forward(xA, xB) -> logits tensor.
    Same parameter names / registration order / default initialisation (kaiming fan_out for Conv2d, BN 1/0,
    SNUNet.py:108-113) as the reference."""

    ARCH = "snunet"

    def __init__(self, in_ch=3, out_ch=1, dtype: Optional[str] = None):
        super().__init__(in_ch, out_ch, dtype)
        n1 = 32
        f = [n1, n1 * 2, n1 * 4, n1 * 8, n1 * 16]
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv0_0 = _NestedBlockHolder(in_ch, f[0], f[0])
        self.conv1_0 = _NestedBlockHolder(f[0], f[1], f[1])
        self.Up1_0 = _UpHolder(f[1])
        self.conv2_0 = _NestedBlockHolder(f[1], f[2], f[2])
        self.Up2_0 = _UpHolder(f[2])
        self.conv3_0 = _NestedBlockHolder(f[2], f[3], f[3])
        self.Up3_0 = _UpHolder(f[3])
        self.conv4_0 = _NestedBlockHolder(f[3], f[4], f[4])
        self.Up4_0 = _UpHolder(f[4])
        self.conv0_1 = _NestedBlockHolder(f[0] * 2 + f[1], f[0], f[0])
        self.conv1_1 = _NestedBlockHolder(f[1] * 2 + f[2], f[1], f[1])
        self.Up1_1 = _UpHolder(f[1])
        self.conv2_1 = _NestedBlockHolder(f[2] * 2 + f[3], f[2], f[2])
        self.Up2_1 = _UpHolder(f[2])
        self.conv3_1 = _NestedBlockHolder(f[3] * 2 + f[4], f[3], f[3])
        self.Up3_1 = _UpHolder(f[3])
        self.conv0_2 = _NestedBlockHolder(f[0] * 3 + f[1], f[0], f[0])
        self.conv1_2 = _NestedBlockHolder(f[1] * 3 + f[2], f[1], f[1])
        self.Up1_2 = _UpHolder(f[1])
        self.conv2_2 = _NestedBlockHolder(f[2] * 3 + f[3], f[2], f[2])
        self.Up2_2 = _UpHolder(f[2])
        self.conv0_3 = _NestedBlockHolder(f[0] * 4 + f[1], f[0], f[0])
        self.conv1_3 = _NestedBlockHolder(f[1] * 4 + f[2], f[1], f[1])
        self.Up1_3 = _UpHolder(f[1])
        self.conv0_4 = _NestedBlockHolder(f[0] * 5 + f[1], f[0], f[0])
        self.ca = _ChannelAttentionHolder(f[0] * 4, ratio=16)
        self.ca1 = _ChannelAttentionHolder(f[0], ratio=16 // 4)
        self.conv_final = nn.Conv2d(f[0] * 4, out_ch, kernel_size=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._check_layout()

    def __deepcopy__(self, memo):
        new = type(self)(self._engine.in_ch, self._engine.label_ch, self._engine.dtype)
        new.load_state_dict({k: v.detach().clone() for k, v in self.state_dict().items()})
        new.train(self.training)
        if self._flat_params is not None:
            new.to(self._flat_params.device)
        return new
