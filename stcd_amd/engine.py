"""Python handle over the C-ABI engine (include/stcd_hip.h).

torch is plumbing here: it owns device memory (parameters, workspace, outputs) and the HIP stream the
engine enqueues on.  All arithmetic happens in libstcd_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import _lib


@dataclass
class ParamInfo:
    name: str
    shape: Tuple[int, ...]
    offset: int
    numel: int


@dataclass
class BnInfo:
    name: str
    channels: int
    calls_per_forward: int
    offset: int


@dataclass
class DropoutInfo:
    name: str
    rows: int
    channels: int
    offset: int


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    def __init__(self, arch: str, in_ch: int, label_ch: int, dtype: str = "bf16", cf_config: Optional[dict] = None):
        self._l = _lib.lib()
        self.arch, self.in_ch, self.label_ch, self.dtype = arch, in_ch, label_ch, dtype
        self.cf_config = dict(cf_config) if cf_config else None
        h = C.c_void_p()
        if arch == "changeformer":
            # ChangeFormerV6.__init__ values (/root/reference/models/ChangeFormer.py:1671-1691) unless overridden
            cfg = _lib.CfConfig()
            _lib.check(self._l.stcd_cf_default_config(C.byref(cfg)))
            cfg.in_ch, cfg.out_ch = in_ch, label_ch
            for k, v in (self.cf_config or {}).items():
                if k in ("embed_dims", "depths", "num_heads", "sr_ratios"):
                    arr = getattr(cfg, k)
                    for i in range(4):
                        arr[i] = int(v[i])
                elif k in ("drop_rate", "attn_drop", "drop_path_rate", "diff_drop"):
                    setattr(cfg, k, float(v))
                elif k in ("mlp_ratio", "embedding_dim", "patch1", "patch"):
                    setattr(cfg, k, int(v))
                else:
                    raise _lib.StcdError(f"unknown ChangeFormer option {k!r}")
            _lib.check(self._l.stcd_create_changeformer(C.byref(cfg), _lib.DTYPE_IDS[dtype], C.byref(h)))
        else:
            _lib.check(self._l.stcd_create(_lib.ARCH_IDS[arch], in_ch, label_ch, _lib.DTYPE_IDS[dtype], C.byref(h)))
        self._h = h
        self.params: List[ParamInfo] = []
        ti = _lib.TensorInfo()
        for i in range(self._l.stcd_num_params(h)):
            _lib.check(self._l.stcd_param_info(h, i, C.byref(ti)))
            self.params.append(ParamInfo(ti.name.decode(), tuple(ti.shape[k] for k in range(ti.ndim)), ti.offset, ti.numel))
        self.param_floats = self._l.stcd_param_floats(h)
        self.bns: List[BnInfo] = []
        bi = _lib.BnInfo()
        for i in range(self._l.stcd_num_bn(h)):
            _lib.check(self._l.stcd_bn_info_get(h, i, C.byref(bi)))
            self.bns.append(BnInfo(bi.name.decode(), bi.channels, bi.calls_per_forward, bi.offset))
        self.bn_floats = self._l.stcd_bn_floats(h)
        b, e = C.c_int64(), C.c_int64()
        _lib.check(self._l.stcd_grad_stage_range(h, 0, C.byref(b), C.byref(e)))
        self.stage0_range = (b.value, e.value)     # decoder gradients: final after backward stage 0
        _lib.check(self._l.stcd_grad_stage_range(h, 1, C.byref(b), C.byref(e)))
        self.stage1_range = (b.value, e.value)
        self.shape: Optional[Tuple[int, int, int]] = None
        self.dropouts: List[DropoutInfo] = []
        self.dropout_floats = 0
        self.workspace: Optional[torch.Tensor] = None
        self.ticket = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._l.stcd_destroy(h)

    # pickling / deepcopy re-create the native handle (it is process-local)
    def __getstate__(self):
        return {"arch": self.arch, "in_ch": self.in_ch, "label_ch": self.label_ch, "dtype": self.dtype, "cf_config": self.cf_config}

    def __setstate__(self, st):
        self.__init__(st["arch"], st["in_ch"], st["label_ch"], st["dtype"], st.get("cf_config"))

    def __deepcopy__(self, memo):
        return Engine(self.arch, self.in_ch, self.label_ch, self.dtype, self.cf_config)

    def set_dropout_p(self, p: float):
        _lib.check(self._l.stcd_set_dropout_p(self._h, C.c_float(p)))

    def configure(self, batch: int, height: int, width: int, device: torch.device):
        if self.shape == (batch, height, width) and self.workspace is not None and self.workspace.device == device:
            return
        _lib.check(self._l.stcd_configure(self._h, batch, height, width))
        self.shape = (batch, height, width)
        self.dropouts = []
        di = _lib.DropoutInfo()
        for i in range(self._l.stcd_num_dropout(self._h)):
            _lib.check(self._l.stcd_dropout_info_get(self._h, i, C.byref(di)))
            self.dropouts.append(DropoutInfo(di.name.decode(), di.rows, di.channels, di.offset))
        self.dropout_floats = self._l.stcd_dropout_floats(self._h)
        self.workspace = None   # release before re-allocating
        self.workspace = torch.empty(self._l.stcd_workspace_bytes(self._h), dtype=torch.uint8, device=device)

    # ------------------------------------------------------------------ ChangeFormer: output layout, dropout sites
    def output_floats(self) -> int:
        return int(self._l.stcd_output_floats(self._h))

    def cf_outputs(self):
        """-> [(offset, height, width)] of the five maps [p_c4, p_c3, p_c2, p_c1, cp] inside the flat output buffer."""
        out = []
        for i in range(5):
            off, hh, ww = C.c_int64(), C.c_int(), C.c_int()
            _lib.check(self._l.stcd_cf_output_info(self._h, i, C.byref(off), C.byref(hh), C.byref(ww)))
            out.append((off.value, hh.value, ww.value))
        return out

    def cf_sites(self):
        """-> [(name, dims, p)] of every Dropout / DropPath site in the engine's order (see stcd_cf_site in the header)."""
        st = _lib.CfSite()
        out = []
        for i in range(self._l.stcd_cf_num_sites(self._h)):
            _lib.check(self._l.stcd_cf_site_get(self._h, i, C.byref(st)))
            out.append((st.name.decode(), tuple(st.dims[k] for k in range(st.ndim)), float(st.p)))
        return out

    def cf_set_drop_rates(self, drop_rate: float, attn_drop: float, diff_drop: float):
        _lib.check(self._l.stcd_cf_set_drop_rates(self._h, C.c_float(drop_rate), C.c_float(attn_drop), C.c_float(diff_drop)))
        self.shape = None

    def cf_set_aux_backward(self, on: bool):
        """ChangeFormer: plan and run the backward of the four auxiliary prediction heads too (multi_scale_train); takes effect at
        the next configure (the workspace plan changes)."""
        _lib.check(self._l.stcd_cf_set_aux_backward(self._h, 1 if on else 0))
        self.shape = None

    def set_wgrad_side(self, on: bool):
        """FC-Siam family: run the decoder's weight gradients on the engine's side stream beside the encoder's backward chain
        (single-call backward only).  Off for data-parallel training, whose staged backward keeps everything on one stream; the
        plan is rebuilt by the next forward."""
        _lib.check(self._l.stcd_set_wgrad_side(self._h, 1 if on else 0))
        self.shape = None

    def pack_masks(self, masks: dict, device) -> torch.Tensor:
        """name -> [rows, C] tensors (oracle convention) into the engine's flat mask buffer."""
        flat = torch.empty(self.dropout_floats, dtype=torch.float32, device=device)
        for d in self.dropouts:
            m = masks[d.name]
            assert tuple(m.shape) == (d.rows, d.channels), (d.name, tuple(m.shape), d.rows, d.channels)
            flat[d.offset:d.offset + d.rows * d.channels] = m.reshape(-1).to(device)
        return flat

    def forward(self, x1, x2, flat_params, flat_bn, logits, training: bool, masks: Optional[torch.Tensor] = None,
                seed: int = 0):
        self.ticket += 1
        _lib.check(self._l.stcd_forward(self._h, _ptr(x1), _ptr(x2), _ptr(flat_params), _ptr(flat_bn), _ptr(masks),
                                        C.c_uint64(seed & (2 ** 64 - 1)), int(training), _ptr(logits),
                                        _ptr(self.workspace), _stream()))
        return self.ticket

    def backward(self, grad_logits, flat_params, flat_grads, stage: int = -1):
        _lib.check(self._l.stcd_backward(self._h, _ptr(grad_logits), _ptr(flat_params), _ptr(flat_grads),
                                         _ptr(self.workspace), stage, _stream()))

    def set_weights_tag(self, tag: int):
        """Non-zero: the caller vouches that the parameters do not change under this tag (the filter repack is skipped)."""
        _lib.check(self._l.stcd_set_weights_tag(self._h, C.c_uint64(tag & (2 ** 64 - 1))))

    # ------------------------------------------------------------------ test introspection
    def set_debug(self, flags: int):
        """bit 0: every layer keeps its input gradient in a buffer of its own (see include/stcd_hip.h)."""
        _lib.check(self._l.stcd_set_debug(self._h, int(flags)))
        self.shape = None

    def ws_tensors(self):
        """-> {name: NHWC view [n, h, w, c] of the live workspace} for the current configuration (SegCD only)."""
        out = {}
        wt = _lib.WsTensor()
        for i in range(self._l.stcd_ws_tensor_count(self._h)):
            _lib.check(self._l.stcd_ws_tensor_get(self._h, i, C.byref(wt)))
            dt = torch.bfloat16 if wt.dtype == _lib.DTYPE_IDS["bf16"] else torch.float32
            es = 2 if dt == torch.bfloat16 else 4
            flat = self.workspace[wt.offset_bytes:wt.offset_bytes + wt.n * wt.h * wt.w * wt.ld * es].view(dt)
            out[wt.name.decode()] = flat.view(wt.n, wt.h, wt.w, wt.ld)[..., :wt.c]
        return out

    # ------------------------------------------------------------------ measurement aid (bench.py)
    PROFILE_CLASSES = ("conv", "wgrad", "bn_stats", "bn_act", "bn_bwd_reduce", "bn_bwd_apply", "pool_fuse_bwd", "pack")

    def profile_enable(self, on: bool):
        _lib.check(self._l.stcd_profile_enable(self._h, int(on)))

    def profile_read(self):
        """-> {class: dict(ms, launches, flops, bytes)} for the launches recorded since profile_enable(True)."""
        out = {}
        for k, name in enumerate(self.PROFILE_CLASSES):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _lib.check(self._l.stcd_profile_read(self._h, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out

    def profile_kernels(self):
        """-> {kernel name (as rocprofv3 prints it, substring): dict(ms, launches, flops, bytes)}."""
        out = {}
        buf = C.create_string_buffer(96)
        for i in range(self._l.stcd_profile_num_kernels(self._h)):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _lib.check(self._l.stcd_profile_kernel(self._h, i, buf, 96, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[buf.value.decode()] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out
