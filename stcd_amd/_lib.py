"""ctypes binding of the C ABI in include/stcd_hip.h (stcd_amd/libstcd_hip.so).

There is no fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

# torch wheels bundle their own libamdhip64; it must be the HIP runtime this process uses (device memory and
# streams come from torch).  Importing torch first makes the dynamic linker resolve our library's
# libamdhip64 dependency to that already-loaded runtime instead of a second copy from /opt/rocm.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# STCD_LIB_PATH: development aid for A/B runs of two builds of the library in one GPU session (tools/ab_variants.sh); the product
# loads stcd_amd/libstcd_hip.so
LIB_PATH = os.environ.get("STCD_LIB_PATH") or os.path.join(_HERE, "libstcd_hip.so")

ARCH_DIFF, ARCH_CONC, ARCH_SUB, ARCH_SNUNET, ARCH_SEGCD = 0, 1, 2, 3, 4
ARCH_SEGCD_R18, ARCH_SEGCD_R34, ARCH_SEGCD_R101, ARCH_SEGCD_R152 = 5, 6, 7, 8
ARCH_FCEF = 9              # Unet (FC-EF)
ARCH_XCONC = 10            # SiamUnet_cross_conc
ARCH_UNETSEG = 16          # + 0..4: resnet50, resnet18, resnet34, resnet101, resnet152
ARCH_FFCTLCD = 32          # + 0..4, same order
ARCH_CHANGEFORMER = 64     # ChangeFormerV6
DTYPE_F32, DTYPE_BF16 = 0, 1
ARCH_IDS = {"diff": ARCH_DIFF, "conc": ARCH_CONC, "sub": ARCH_SUB, "fcef": ARCH_FCEF, "xconc": ARCH_XCONC, "snunet": ARCH_SNUNET, "segcd": ARCH_SEGCD,
            "segcd_resnet50": ARCH_SEGCD, "segcd_resnet18": ARCH_SEGCD_R18, "segcd_resnet34": ARCH_SEGCD_R34,
            "segcd_resnet101": ARCH_SEGCD_R101, "segcd_resnet152": ARCH_SEGCD_R152,
            "unetseg_resnet50": ARCH_UNETSEG, "unetseg_resnet18": ARCH_UNETSEG + 1, "unetseg_resnet34": ARCH_UNETSEG + 2,
            "unetseg_resnet101": ARCH_UNETSEG + 3, "unetseg_resnet152": ARCH_UNETSEG + 4,
            "ffctlcd_resnet50": ARCH_FFCTLCD, "ffctlcd_resnet18": ARCH_FFCTLCD + 1, "ffctlcd_resnet34": ARCH_FFCTLCD + 2,
            "ffctlcd_resnet101": ARCH_FFCTLCD + 3, "ffctlcd_resnet152": ARCH_FFCTLCD + 4,
            "changeformer": ARCH_CHANGEFORMER}
DTYPE_IDS = {"fp32": DTYPE_F32, "f32": DTYPE_F32, "bf16": DTYPE_BF16}


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("ndim", C.c_int32), ("shape", C.c_int64 * 4), ("offset", C.c_int64),
                ("numel", C.c_int64)]


class BnInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("channels", C.c_int32), ("calls_per_forward", C.c_int32),
                ("offset", C.c_int64)]


class DropoutInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("rows", C.c_int32), ("channels", C.c_int32), ("offset", C.c_int64)]


class ConvGeom(C.Structure):
    _fields_ = [("n", C.c_int32), ("hi", C.c_int32), ("wi", C.c_int32), ("ci", C.c_int32), ("ldi", C.c_int32),
                ("hm", C.c_int32), ("wm", C.c_int32), ("in_stride", C.c_int32),
                ("ho", C.c_int32), ("wo", C.c_int32), ("out_stride", C.c_int32), ("oy0", C.c_int32), ("ox0", C.c_int32),
                ("co", C.c_int32), ("ldo", C.c_int32), ("ntaps", C.c_int32),
                ("dy", C.c_int8 * 9), ("dx", C.c_int8 * 9), ("pad_", C.c_int8 * 2)]


class WsTensor(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset_bytes", C.c_int64), ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int),
                ("ld", C.c_int), ("dtype", C.c_int)]


class CfConfig(C.Structure):
    _fields_ = [("in_ch", C.c_int32), ("out_ch", C.c_int32), ("embed_dims", C.c_int32 * 4), ("depths", C.c_int32 * 4),
                ("num_heads", C.c_int32 * 4), ("sr_ratios", C.c_int32 * 4), ("mlp_ratio", C.c_int32), ("embedding_dim", C.c_int32),
                ("patch1", C.c_int32), ("patch", C.c_int32), ("drop_rate", C.c_float), ("attn_drop", C.c_float),
                ("drop_path_rate", C.c_float), ("diff_drop", C.c_float)]


class CfSite(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("ndim", C.c_int32), ("dims", C.c_int32 * 4), ("p", C.c_float)]


class MapGeom(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("groups", C.c_int32)]


_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_PROTOS = {
    "stcd_last_error": (C.c_char_p, []),
    "stcd_abi_version": (_i, []),
    "stcd_create": (_i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    "stcd_destroy": (None, [_vp]),
    "stcd_cf_default_config": (_i, [C.POINTER(CfConfig)]),
    "stcd_create_changeformer": (_i, [C.POINTER(CfConfig), _i, C.POINTER(_vp)]),
    "stcd_output_floats": (_i64, [_vp]),
    "stcd_cf_output_info": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i)]),
    "stcd_cf_num_sites": (_i, [_vp]),
    "stcd_cf_site_get": (_i, [_vp, _i, C.POINTER(CfSite)]),
    "stcd_cf_site_seed": (C.c_uint32, [C.c_uint64, _i]),
    "stcd_cf_set_drop_rates": (_i, [_vp, _f, _f, _f]),
    "stcd_cf_set_aux_backward": (_i, [_vp, _i]),
    "stcd_set_wgrad_side": (_i, [_vp, _i]),
    "stcd_num_params": (_i, [_vp]),
    "stcd_param_info": (_i, [_vp, _i, C.POINTER(TensorInfo)]),
    "stcd_param_floats": (_i64, [_vp]),
    "stcd_num_bn": (_i, [_vp]),
    "stcd_bn_info_get": (_i, [_vp, _i, C.POINTER(BnInfo)]),
    "stcd_bn_floats": (_i64, [_vp]),
    "stcd_configure": (_i, [_vp, _i, _i, _i]),
    "stcd_workspace_bytes": (_i64, [_vp]),
    "stcd_num_dropout": (_i, [_vp]),
    "stcd_dropout_info_get": (_i, [_vp, _i, C.POINTER(DropoutInfo)]),
    "stcd_dropout_floats": (_i64, [_vp]),
    "stcd_set_dropout_p": (_i, [_vp, _f]),
    "stcd_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _i, _vp, _vp, _vp]),
    "stcd_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "stcd_grad_stage_range": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "stcd_set_weights_tag": (_i, [_vp, C.c_uint64]),
    "stcd_set_debug": (_i, [_vp, _i]),
    "stcd_ws_tensor_count": (_i, [_vp]),
    "stcd_ws_tensor_get": (_i, [_vp, _i, C.POINTER(WsTensor)]),
    "stcd_profile_enable": (_i, [_vp, _i]),
    "stcd_profile_read": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "stcd_profile_num_kernels": (_i, [_vp]),
    "stcd_profile_kernel": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "stcd_loss_scratch_bytes": (_i64, []),
    "stcd_loss_ce": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "stcd_loss_bce_dice": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "stcd_loss_contrastive": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "stcd_confusion_update": (_i, [_vp, _vp, _i, _i, _i64, _vp, _vp]),
    "stcd_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _d, _d, _d, _d, _d, _i, _vp]),
    "stcd_adam_hyper": (_i, [_i64, _d, _d, _d, _d, _d, C.POINTER(_f)]),
    "stcd_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _i, _vp]),
    "stcd_pseudo_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _vp, _vp, _vp, _vp, _vp]),
    "stcd_augment_scratch_bytes": (_i64, [_i, _i, _i]),
    "stcd_augment": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _vp, _i64, _vp]),
    "stcd_op_conv": (_i, [_i, _i, C.POINTER(ConvGeom), _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "stcd_op_wgrad": (_i, [_i, _i, C.POINTER(ConvGeom), _vp, _vp, _vp, _vp, _i64, _vp]),
    "stcd_op_scratch_bytes": (_i64, [C.POINTER(ConvGeom)]),
    "stcd_op_ew_scratch_bytes": (_i64, [C.POINTER(MapGeom)]),
    "stcd_op_bn_act": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp]),
    "stcd_op_bn_act_pair": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp]),
    "stcd_op_bn_act_bwd": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp]),
    "stcd_op_maxpool": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp]),
    "stcd_op_maxpool_bwd": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "stcd_op_maxpool3": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp, _vp]),
    "stcd_op_maxpool3_bwd": (_i, [_i, C.POINTER(MapGeom), _vp, _vp, _i, _vp, _i, _vp]),
    "stcd_op_pairdw_scratch_bytes": (_i64, [C.POINTER(MapGeom)]),
    "stcd_op_pairdw": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "stcd_op_pairdw_bwd": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i64, _vp]),
    "stcd_op_fuse": (_i, [_i, _i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp]),
    "stcd_op_fuse_bwd": (_i, [_i, _i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp, _i, _vp]),
    "stcd_op_rep_pad": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _i, _i, _vp]),
    "stcd_op_rep_pad_bwd": (_i, [_i, C.POINTER(MapGeom), _vp, _i, _i, _i, _vp]),
    "stcd_op_skip_bwd": (_i, [_i, _i, C.POINTER(MapGeom), _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp]),
    "stcd_op_cf_scratch_bytes": (_i64, [_i64, _i, _i, _i, _i]),
    "stcd_op_cf_im2col": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "stcd_op_cf_col2im": (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "stcd_op_cf_layernorm": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    "stcd_op_cf_layernorm_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "stcd_op_cf_colsum": (_i, [_i, _vp, _i64, _i, _vp, _vp, _vp]),
    "stcd_op_cf_attention": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, C.c_uint64, _vp]),
    "stcd_op_cf_attention_scratch_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "stcd_op_cf_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, C.c_uint64, _vp]),
    "stcd_op_cf_dwgelu": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, C.c_uint64, _vp]),
    "stcd_op_cf_dwgelu_scratch_bytes": (_i64, [_i, _i, _i, _i]),
    "stcd_op_cf_dwgelu_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, C.c_uint64, _vp]),
    "stcd_op_cf_resid_drop": (_i, [_i, _vp, _vp, _vp, _i, _i64, _i, _f, _f, C.c_uint64, _i, _vp]),
    "stcd_op_cf_bilinear": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "stcd_op_cf_elementwise": (_i, [_i, _i, _vp, _vp, _vp, _i64, _i, _vp, _f, _f, _f, C.c_uint64, _vp]),
    "stcd_op_cf_prelu_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
}
EXPORTS = tuple(_PROTOS)

_lib = None


class StcdError(RuntimeError):
    pass


def lib():
    """Load libstcd_hip.so (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StcdError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C stcd_amd/csrc`.  The engine has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.stcd_abi_version() != 2:
            raise StcdError("libstcd_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise StcdError(lib().stcd_last_error().decode("utf-8", "replace"))
