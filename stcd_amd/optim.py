"""Fused Adam / AdamW over the engine's flat parameter buffer: ONE kernel per ``step()``.

Drop-in for the optimizers the reference builds (``optim.Adam(net_G.parameters(), lr, weight_decay=0)`` /
``optim.AdamW(..., betas=(0.9, 0.999), weight_decay=0.01)``: /root/reference/models/trainer.py:46-50;
``torch.optim.Adam(model.parameters(), lr=0.001, betas=(0.9, 0.999))``: /root/reference/train_pse_cd.py:431).
It IS a ``torch.optim.Optimizer`` (one param group over the module's parameters), so ``get_scheduler`` /
``Poly`` and ``state_dict`` checkpoints work unchanged; the update itself is ``stcd_adam_step`` in
``include/stcd_hip.h`` applied to ``model._flat_params`` / ``model._flat_grads`` (the tensors the reference optimizer
would walk one by one are views of those two buffers).  amsgrad / maximize / foreach are not offered: the reference
never sets them.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import StcdError


class _FlatAdamBase(torch.optim.Optimizer):
    DECOUPLED = False

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        from .modules import HipChangeDetector
        inner = model.module if hasattr(model, "module") and not isinstance(model, HipChangeDetector) else model
        if not isinstance(inner, HipChangeDetector):
            raise StcdError("the flat optimizer drives a stcd_amd engine module (SiamUnet_* / SNUNet_ECAM)")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        self._model = inner
        super().__init__(list(inner.parameters()), dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._exp_avg: Optional[torch.Tensor] = None
        self._exp_avg_sq: Optional[torch.Tensor] = None
        self._step = 0

    # ---- flat state tied to the model's current flat buffers
    def _ensure_state(self):
        m = self._model
        dev = m._flat_params.device if getattr(m, "_flat_params", None) is not None else next(m.parameters()).device
        if dev.type != "cuda":
            raise StcdError("move the model to the GPU before stepping / loading the flat optimizer (no CPU fallback)")
        m._ensure_flat(dev)                    # no-op once the flat views exist (also re-flattens after .to())
        fp = m._flat_params
        if self._exp_avg is None or self._exp_avg.device != fp.device or self._exp_avg.numel() != fp.numel():
            old_a, old_s = self._exp_avg, self._exp_avg_sq
            self._exp_avg = torch.zeros_like(fp)
            self._exp_avg_sq = torch.zeros_like(fp)
            if old_a is not None and old_a.numel() == fp.numel():
                self._exp_avg.copy_(old_a)
                self._exp_avg_sq.copy_(old_s)

    def _gather_grads(self) -> torch.Tensor:
        """The engine leaves p.grad as views of model._flat_grads; anything else (accumulated / user-made grads) is
        copied into place first."""
        m = self._model
        fg = m._flat_grads
        for p, info in zip(m._live_params() or m.parameters(), m._engine.params):
            if p.grad is None:
                raise StcdError("step() without gradients: call backward() first")
            if p.grad.data_ptr() != fg.data_ptr() + 4 * info.offset:
                fg[info.offset:info.offset + info.numel].view(info.shape).copy_(p.grad)
        return fg

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure_state()
        g = self.param_groups[0]
        m = self._model
        fg = self._gather_grads()
        fp = m._flat_params
        self._step += 1
        if hasattr(m, "_weights_changed"):
            m._weights_changed()            # a step inside model.frozen_weights(): the vouch is void, the next forward repacks
        with torch.cuda.device(fp.device):
            _lib.check(_lib.lib().stcd_adam_step(
                C.c_void_p(fp.data_ptr()), C.c_void_p(fg.data_ptr()), C.c_void_p(self._exp_avg.data_ptr()),
                C.c_void_p(self._exp_avg_sq.data_ptr()), fp.numel(), self._step, float(g["lr"]), float(g["betas"][0]),
                float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), 1 if self.DECOUPLED else 0,
                C.c_void_p(torch.cuda.current_stream(fp.device).cuda_stream)))
        return loss

    # ---- captured steps (stcd_amd.train_loop.GraphedTrainStep): the step-dependent scalars live in DEVICE memory, so ONE captured
    #      launch serves every replay.  The host writes them into a ring of pinned slots and enqueues the slot's copy in front of the
    #      replay (stream order); a slot is rewritten only after the copy that read it has completed (event), so the host may run
    #      several steps ahead of the GPU.
    _SLOTS = 8

    def prepare_graph_step(self):
        """Host side of a graphed step (call BEFORE graph.replay()): advance the step count, write this step's scalars into the next
        pinned slot and enqueue its copy to the device buffer the captured kernel reads."""
        self._ensure_state()
        g = self.param_groups[0]
        dev = self._model._flat_params.device
        if getattr(self, "_hyper_pin", None) is None:
            self._hyper_pin = [torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(self._SLOTS)]
            self._hyper_ev = [None] * self._SLOTS
            self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=dev)
        self._step += 1
        k = self._step % self._SLOTS
        if self._hyper_ev[k] is not None:
            self._hyper_ev[k].synchronize()
        arr = (C.c_float * 8)()
        _lib.check(_lib.lib().stcd_adam_hyper(self._step, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                              float(g["weight_decay"]), arr))
        self._hyper_pin[k].copy_(torch.tensor(list(arr), dtype=torch.float32))
        self._hyper_dev.copy_(self._hyper_pin[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._hyper_ev[k] = ev

    @torch.no_grad()
    def step_graph(self):
        """Device side (call INSIDE the capture): one launch that reads the step's scalars from device memory."""
        m = self._model
        fg, fp = self._gather_grads(), m._flat_params
        if hasattr(m, "_weights_changed"):
            m._weights_changed()
        with torch.cuda.device(fp.device):
            _lib.check(_lib.lib().stcd_adam_step_dev(
                C.c_void_p(fp.data_ptr()), C.c_void_p(fg.data_ptr()), C.c_void_p(self._exp_avg.data_ptr()),
                C.c_void_p(self._exp_avg_sq.data_ptr()), fp.numel(), C.c_void_p(self._hyper_dev.data_ptr()), 1 if self.DECOUPLED else 0,
                C.c_void_p(torch.cuda.current_stream(fp.device).cuda_stream)))

    # ---- checkpoints: torch's per-parameter layout ({'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups'})
    def state_dict(self):
        sd = super().state_dict()
        state = {}
        if self._exp_avg is not None:
            for i, info in enumerate(self._model._engine.params):
                sl = slice(info.offset, info.offset + info.numel)
                state[i] = {"step": torch.tensor(float(self._step)),
                            "exp_avg": self._exp_avg[sl].view(info.shape).clone(),
                            "exp_avg_sq": self._exp_avg_sq[sl].view(info.shape).clone()}
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        super().load_state_dict({"state": {}, "param_groups": sd["param_groups"]})
        state = sd.get("state", {})
        if state:
            self._ensure_state()
            for i, info in enumerate(self._model._engine.params):
                st = state[i] if i in state else state[str(i)]
                sl = slice(info.offset, info.offset + info.numel)
                self._exp_avg[sl].view(info.shape).copy_(st["exp_avg"])
                self._exp_avg_sq[sl].view(info.shape).copy_(st["exp_avg_sq"])
                self._step = int(float(st["step"]))


class FlatAdam(_FlatAdamBase):
    """torch.optim.Adam semantics (weight decay added to the gradient)."""
    DECOUPLED = False


class FlatAdamW(_FlatAdamBase):
    """torch.optim.AdamW semantics (decoupled weight decay)."""
    DECOUPLED = True

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        super().__init__(model, lr, betas, eps, weight_decay)
