"""Loss functions of the path, same names and argument meaning as the reference
(/root/reference/models/losses.py:6-21 ``cross_entropy``, :24-34 ``cd_loss``;
/root/reference/train_pse_cd.py:436-462 ``Dice`` / ``BCE_DICE``), computed by the fused HIP kernels of
libstcd_hip.so (forward value and gradient in one pass; the autograd node only scales by the incoming
gradient)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import StcdError

_scratch = {}


def _scratch_for(device):
    s = _scratch.get(device)
    if s is None:
        s = torch.empty(_lib.lib().stcd_loss_scratch_bytes(), dtype=torch.uint8, device=device)
        _scratch[device] = s
    return s


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t, what):
    if not t.is_cuda:
        raise StcdError(f"{what}: the HIP loss kernels need tensors on the GPU; there is no CPU fallback")


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        B, Cn = logits.shape[:2]
        hw = logits.numel() // (B * Cn)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        need_grad = logits.requires_grad
        dl = torch.empty_like(logits) if need_grad else None
        with torch.cuda.device(logits.device):
            _lib.check(_lib.lib().stcd_loss_ce(_p(logits), _p(target), B, Cn, hw, ignore_index, _p(loss),
                                               _p(dl) if need_grad else None, _p(_scratch_for(logits.device)), _stream()))
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.dl * g, None, None


def cross_entropy(input, target, weight=None, reduction="mean", ignore_index=255):
    """logSoftmax_with_loss: input N*C*H*W, target N*1*H*W or N*H*W (any numeric dtype)."""
    if weight is not None or reduction != "mean":
        raise NotImplementedError("only the reference's call form (weight=None, reduction='mean') is implemented")
    _need_cuda(input, "cross_entropy")
    target = target.long()
    if target.dim() == 4:
        target = torch.squeeze(target, dim=1)
    if input.shape[-1] != target.shape[-1]:
        input = F.interpolate(input, size=target.shape[1:], mode="bilinear", align_corners=True)
    return _CrossEntropyFn.apply(input.contiguous().float(), target.contiguous(), int(ignore_index))


class _BceDiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, from_logits):
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        need_grad = x.requires_grad
        dx = torch.empty_like(x) if need_grad else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().stcd_loss_bce_dice(_p(x), _p(target), x.numel(), int(from_logits), _p(loss),
                                                     _p(dx) if need_grad else None, _p(_scratch_for(x.device)), _stream()))
        ctx.dx = dx
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.dx * g, None, None


def cd_loss(input, target):
    """BCE(mean) + Dice(smooth=1) on PROBABILITIES, as the reference calls it."""
    _need_cuda(input, "cd_loss")
    return _BceDiceFn.apply(input.contiguous().float(), target.contiguous().float(), False)


def bce_dice_with_logits(logits, target):
    """cd_loss(sigmoid(logits), target) with the sigmoid fused into the kernel (train_pse_cd.py:227-228)."""
    _need_cuda(logits, "bce_dice_with_logits")
    return _BceDiceFn.apply(logits.contiguous().float(), target.contiguous().float(), True)


class _ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, cd_label, pse_label):
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        need_grad = pred.requires_grad
        dp = torch.empty_like(pred) if need_grad else None
        with torch.cuda.device(pred.device):
            _lib.check(_lib.lib().stcd_loss_contrastive(_p(pred), _p(cd_label), _p(pse_label), cd_label.numel(), _p(loss),
                                                        _p(dp) if need_grad else None, _p(_scratch_for(pred.device)), _stream()))
        ctx.dp = dp
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.dp * g, None, None


def contrastive_loss(pred, cd_label, pse_label, img_name=None):
    """/root/reference/train_stcd.py:334-385, same arguments (``img_name`` only fed the reference's commented-out
    visualisation): ``pred`` = change PROBABILITIES of cat(real pairs, pseudo pairs) along the batch axis, the labels of
    the two halves [b,1,H,W]; masked MSE between the halves where the labels agree plus masked MSE against the inverted
    real prediction where they differ.  One fused HIP pass for value and gradient (both halves receive gradient)."""
    _need_cuda(pred, "contrastive_loss")
    b = cd_label.shape[0]
    if pred.shape[0] != 2 * b or cd_label.shape != pse_label.shape or pred[:b].shape != cd_label.shape:
        raise StcdError(f"contrastive_loss: pred {tuple(pred.shape)} must hold 2 x the label batch {tuple(cd_label.shape)}")
    return _ContrastiveFn.apply(pred.contiguous().float(), cd_label.contiguous().long(), pse_label.contiguous().long())


class Dice(nn.Module):
    """train_pse_cd.py:436-447: 1 - (2*sum(p*t) + 1) / (sum(p) + sum(t) + 1) on probabilities."""

    def forward(self, pred, target):
        _need_cuda(pred, "Dice")
        return _BceDiceFn.apply(pred.contiguous().float(), target.contiguous().float(), 2)


class BCE_DICE(nn.Module):
    """train_pse_cd.py:451-462: forward(pmask = sigmoid output, rmask = {0,1} target)."""

    def forward(self, pmask, rmask):
        return cd_loss(pmask, rmask)
