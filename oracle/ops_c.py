"""ctypes front-end of the plain-C per-op oracle (oracle/ops_ref.c).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  numpy in, numpy out, NCHW fp32.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libstcd_oracle.so")
_lib = None
_fp = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ops_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.ref_ce_fwd_bwd.restype = C.c_double
        _lib.ref_bce_dice_fwd_bwd.restype = C.c_double
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


def conv2d_fwd(x, w, b, pad):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    n, ci, h, wd = x.shape
    co, _, k, _ = w.shape
    y = np.empty((n, co, h + 2 * pad - k + 1, wd + 2 * pad - k + 1), np.float32)
    lib().ref_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), n, ci, h, wd, co, k, pad)
    return y


def conv2d_bwd(x, w, gy, pad):
    x, w, gy = _f(x), _f(w), _f(gy)
    n, ci, h, wd = x.shape
    co, _, k, _ = w.shape
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty(co, np.float32)
    lib().ref_conv2d_bwd(_p(x), _p(w), _p(gy), _p(dx), _p(dw), _p(db), n, ci, h, wd, co, k, pad)
    return dx, dw, db


def convT2d_fwd(x, w, b, stride, pad, opad):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    n, ci, h, wd = x.shape
    _, co, k, _ = w.shape
    ho, wo = (h - 1) * stride - 2 * pad + k + opad, (wd - 1) * stride - 2 * pad + k + opad
    y = np.empty((n, co, ho, wo), np.float32)
    lib().ref_convT2d_fwd(_p(x), _p(w), _p(b), _p(y), n, ci, h, wd, co, k, stride, pad, opad)
    return y


def convT2d_bwd(x, w, gy, stride, pad, opad):
    x, w, gy = _f(x), _f(w), _f(gy)
    n, ci, h, wd = x.shape
    _, co, k, _ = w.shape
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty(co, np.float32)
    lib().ref_convT2d_bwd(_p(x), _p(w), _p(gy), _p(dx), _p(dw), _p(db), n, ci, h, wd, co, k, stride, pad, opad)
    return dx, dw, db


def bn_train_fwd(x, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5):
    """-> y, save_mean, save_invstd, new_rmean, new_rvar"""
    x, gamma, beta = _f(x), _f(gamma), _f(beta)
    rm, rv = _f(rmean).copy(), _f(rvar).copy()
    n, c, h, wd = x.shape
    y, sm, si = np.empty_like(x), np.empty(c, np.float32), np.empty(c, np.float32)
    lib().ref_bn_train_fwd(_p(x), _p(gamma), _p(beta), _p(rm), _p(rv), _p(y), _p(sm), _p(si),
                           n, c, h * wd, C.c_float(momentum), C.c_float(eps))
    return y, sm, si, rm, rv


def bn_eval_fwd(x, gamma, beta, rmean, rvar, eps=1e-5):
    x = _f(x)
    n, c, h, wd = x.shape
    y = np.empty_like(x)
    lib().ref_bn_eval_fwd(_p(x), _p(_f(gamma)), _p(_f(beta)), _p(_f(rmean)), _p(_f(rvar)), _p(y), n, c, h * wd,
                          C.c_float(eps))
    return y


def bn_train_bwd(x, gy, gamma, mean, invstd):
    x, gy = _f(x), _f(gy)
    n, c, h, wd = x.shape
    dx, dg, db = np.empty_like(x), np.empty(c, np.float32), np.empty(c, np.float32)
    lib().ref_bn_train_bwd(_p(x), _p(gy), _p(_f(gamma)), _p(_f(mean)), _p(_f(invstd)), _p(dx), _p(dg), _p(db),
                           n, c, h * wd)
    return dx, dg, db


def maxpool2_fwd(x):
    x = _f(x)
    n, c, h, wd = x.shape
    y = np.empty((n, c, h // 2, wd // 2), np.float32)
    arg = np.empty(y.shape, np.uint8)
    lib().ref_maxpool2_fwd(_p(x), _p(y), arg.ctypes.data_as(C.POINTER(C.c_uint8)), n, c, h, wd)
    return y, arg


def maxpool2_bwd(x, gy):
    x, gy = _f(x), _f(gy)
    n, c, h, wd = x.shape
    dx = np.empty_like(x)
    lib().ref_maxpool2_bwd(_p(x), _p(gy), _p(dx), n, c, h, wd)
    return dx


def rep_pad_fwd(x, H, W):
    x = _f(x)
    n, c, h0, w0 = x.shape
    y = np.empty((n, c, H, W), np.float32)
    lib().ref_rep_pad_fwd(_p(x), _p(y), n, c, h0, w0, H, W)
    return y


def rep_pad_bwd(gy, h0, w0):
    gy = _f(gy)
    n, c, H, W = gy.shape
    dx = np.empty((n, c, h0, w0), np.float32)
    lib().ref_rep_pad_bwd(_p(gy), _p(dx), n, c, h0, w0, H, W)
    return dx


def fuse_fwd(a, b, mode):
    a, b = _f(a), _f(b)
    y = np.empty_like(a)
    lib().ref_fuse_fwd(_p(a), _p(b), _p(y), C.c_int64(a.size), mode)
    return y


def fuse_bwd(a, b, g, mode):
    a, b, g = _f(a), _f(b), _f(g)
    da, db = np.empty_like(a), np.empty_like(a)
    lib().ref_fuse_bwd(_p(a), _p(b), _p(g), _p(da), _p(db), C.c_int64(a.size), mode)
    return da, db


def ce_fwd_bwd(logits, target, ignore=255):
    logits = _f(logits)
    target = np.ascontiguousarray(target, dtype=np.int64)
    n, c = logits.shape[:2]
    hw = int(np.prod(logits.shape[2:]))
    dl = np.empty_like(logits)
    loss = lib().ref_ce_fwd_bwd(_p(logits), target.ctypes.data_as(C.POINTER(C.c_int64)), _p(dl), n, c, hw, ignore)
    return loss, dl


def bce_dice_fwd_bwd(logits, target):
    logits, target = _f(logits), _f(target)
    dl = np.empty_like(logits)
    loss = lib().ref_bce_dice_fwd_bwd(_p(logits), _p(target), _p(dl), C.c_int64(logits.size))
    return loss, dl
