"""ctypes binding of libgicap.so (C ABI declared in include/gicap.h).

Fails loudly: there is no fallback implementation.  ``load()`` raises if the
library has not been built (``python __graft_entry__.py`` / ``build.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm must be loaded BEFORE libgicap.so: the library shares device pointers and HIP streams with
# torch, so both have to bind to the one HIP runtime instance that torch ships (loading the system
# libamdhip64 first gives the process two runtimes and "no ROCm-capable device" on the second).
import torch  # noqa: F401

from . import build as _build

MAX_LAYERS = 4
MAX_CONVS = 8
F32, BF16 = 0, 1
LOSS_TYPES = {"standard": 0, "JS": 1, "KL": 2, "hinge": 3, "tv": 4, "rsgan": 5}

c_float_p = C.POINTER(C.c_float)
c_void_p = C.c_void_p


class DecoderDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "L", "V", "E", "H", "NL", "dtype")]


class DecoderParams(C.Structure):
    _fields_ = [("embed", c_void_p), ("w_ih", c_void_p * MAX_LAYERS), ("w_hh", c_void_p * MAX_LAYERS),
                ("b_ih", c_void_p * MAX_LAYERS), ("b_hh", c_void_p * MAX_LAYERS), ("w_out", c_void_p), ("b_out", c_void_p)]


class DecoderGrads(C.Structure):
    _fields_ = DecoderParams._fields_ + [("features", c_void_p)]


class DecoderShadow(C.Structure):
    _fields_ = [("wcat", c_void_p * MAX_LAYERS), ("bsum", c_void_p * MAX_LAYERS), ("wout", c_void_p), ("wcat_t", c_void_p * MAX_LAYERS)]


class DecoderState(C.Structure):
    _fields_ = [("xh", c_void_p * MAX_LAYERS), ("gates", c_void_p * MAX_LAYERS), ("c", c_void_p * MAX_LAYERS),
                ("hout", c_void_p), ("logits", c_void_p), ("gpre", c_void_p), ("part", c_void_p)]


class DecoderSampleOpts(C.Structure):
    _fields_ = [("h0", c_void_p), ("c0", c_void_p), ("force_ids", c_void_p), ("force_len", c_void_p), ("no_state", C.c_int32),
                ("resume_from", c_void_p), ("resume_B", C.c_int32), ("host_active_rows", c_void_p),
                ("dev_scalars", c_void_p), ("seed_slot", C.c_int32)]


STEP_SEEDS = 6


class StepScalars(C.Structure):
    """gic_step_scalars: the per-step scalars a replayed step graph reads from device memory."""
    _fields_ = [("temperature", C.c_float), ("reserved", C.c_uint32), ("seed", C.c_uint64 * STEP_SEEDS)]


class DecoderBwdWs(C.Structure):
    _fields_ = [("dlogits", c_void_p), ("dhout", c_void_p), ("dgates", c_void_p * MAX_LAYERS),
                ("dxh", c_void_p * MAX_LAYERS), ("dc", c_void_p * MAX_LAYERS)]


class AttnDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "L", "V", "E", "H", "C", "P", "A", "dtype")]


class AttnParams(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("embed", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out", "w_f", "b_f", "w_h", "w_a")]


class AttnGrads(C.Structure):
    _fields_ = AttnParams._fields_ + [("features", c_void_p)]


class AttnShadow(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("wcat", "bsum", "wout", "wcat_t", "wf", "wh")]


class AttnState(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("xh", "gates", "c", "hout", "part", "fproj", "alpha", "hproj")]


class AttnBwdWs(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("dlogits", "dhout", "dgates", "dc", "dz", "dalpha", "dh_extra", "dhproj", "dfproj", "dfproj_act", "dwa_rows", "dx")]


class DiscDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("V", C.c_int32), ("De", C.c_int32), ("R", C.c_int32),
                ("nconv", C.c_int32), ("fsize", C.c_int32 * MAX_CONVS), ("nfilt", C.c_int32 * MAX_CONVS),
                ("F", C.c_int32), ("Fp", C.c_int32), ("dtype", C.c_int32), ("drop_p", C.c_float)]


class DiscParams(C.Structure):
    _fields_ = [("emb", c_void_p), ("conv_w", c_void_p * MAX_CONVS), ("conv_b", c_void_p * MAX_CONVS),
                ("hw_w", c_void_p), ("hw_b", c_void_p), ("f2o_w", c_void_p), ("f2o_b", c_void_p),
                ("o2l_w", c_void_p), ("o2l_b", c_void_p)]


class DiscGrads(C.Structure):
    _fields_ = DiscParams._fields_


class DiscShadow(C.Structure):
    _fields_ = [("emb", c_void_p), ("hw_w", c_void_p), ("f2o_w", c_void_p), ("hw_w_t", c_void_p)]


class DiscState(C.Structure):
    _fields_ = [("emb", c_void_p), ("pooled", c_void_p), ("argmax", c_void_p), ("hpre", c_void_p),
                ("keep", c_void_p), ("ydrop", c_void_p), ("feat", c_void_p)]


class DiscBwdWs(C.Structure):
    _fields_ = [("dfeat", c_void_p), ("dh", c_void_p), ("dydrop", c_void_p), ("dpooled", c_void_p), ("demb", c_void_p)]


class BnRunningDesc(C.Structure):
    _fields_ = [("stats", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p), ("count", C.c_float),
                ("momentum", C.c_float), ("C", C.c_int32), ("nrep", C.c_int32)]


_P = C.POINTER
_SIGNATURES = {
    "gic_abi_version": (C.c_int, []),
    "gic_last_error": (C.c_char_p, []),
    "gic_gemm": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                           C.c_int, C.c_int, C.c_int, C.c_int, c_void_p, C.c_int, C.c_float, c_void_p]),
    "gic_cast2d": (C.c_int, [c_void_p, C.c_int, C.c_int64, c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, c_void_p]),
    "gic_decoder_state_bytes": (C.c_int, [_P(DecoderDims), c_void_p]),
    "gic_decoder_bwd_ws_bytes": (C.c_int, [_P(DecoderDims), c_void_p]),
    "gic_disc_state_bytes": (C.c_int, [_P(DiscDims), c_void_p]),
    "gic_disc_bwd_ws_bytes": (C.c_int, [_P(DiscDims), c_void_p]),
    "gic_decoder_prepare": (C.c_int, [_P(DecoderDims), _P(DecoderParams), _P(DecoderShadow), c_void_p]),
    "gic_decoder_sample_fwd": (C.c_int, [_P(DecoderDims), _P(DecoderParams), _P(DecoderShadow), _P(DecoderState), c_void_p,
                                         c_void_p, C.c_uint64, C.c_float, C.c_int, c_void_p, c_void_p, _P(DecoderSampleOpts), c_void_p]),
    "gic_decoder_forward_tf": (C.c_int, [_P(DecoderDims), _P(DecoderParams), _P(DecoderShadow), _P(DecoderState), c_void_p, c_void_p,
                                         c_void_p, C.c_int, c_void_p, C.c_uint64, C.c_float, C.c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "gic_decoder_forward_tf_bwd": (C.c_int, [_P(DecoderDims), _P(DecoderParams), _P(DecoderShadow), _P(DecoderState), _P(DecoderBwdWs),
                                             c_void_p, c_void_p, c_void_p, C.c_int, c_void_p, C.c_float, C.c_int, _P(DecoderGrads), c_void_p]),
    "gic_decoder_sample_bwd": (C.c_int, [_P(DecoderDims), _P(DecoderParams), _P(DecoderShadow), _P(DecoderState),
                                         _P(DecoderBwdWs), c_void_p, c_void_p, c_void_p, C.c_float, C.c_int,
                                         _P(DecoderGrads), C.c_int, c_void_p, c_void_p]),
    "gic_step_scalars_set": (C.c_int, [c_void_p, _P(StepScalars), c_void_p]),
    "gic_debug_decoder_step": (None, [C.c_int]),
    "gic_decoder_fused_rollout_rows": (C.c_int, [_P(DecoderDims), c_void_p]),
    "gic_attn_prepare": (C.c_int, [_P(AttnDims), _P(AttnParams), _P(AttnShadow), c_void_p]),
    "gic_attn_sample_fwd": (C.c_int, [_P(AttnDims), _P(AttnParams), _P(AttnShadow), _P(AttnState), c_void_p, c_void_p, c_void_p, C.c_uint64,
                                      C.c_float, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, c_void_p]),
    "gic_attn_sample_bwd": (C.c_int, [_P(AttnDims), _P(AttnParams), _P(AttnShadow), _P(AttnState), _P(AttnBwdWs), c_void_p, c_void_p, c_void_p,
                                      c_void_p, C.c_float, C.c_int, _P(AttnGrads), c_void_p, c_void_p]),
    "gic_embedding_fwd": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_int32, c_void_p]),
    "gic_embedding_bwd": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int, c_void_p]),
    "gic_disc_prepare": (C.c_int, [_P(DiscDims), _P(DiscParams), _P(DiscShadow), c_void_p]),
    "gic_disc_fwd": (C.c_int, [_P(DiscDims), _P(DiscParams), _P(DiscShadow), _P(DiscState), c_void_p, C.c_int64, c_void_p,
                               C.c_int, c_void_p, C.c_uint64, c_void_p, c_void_p, C.c_int, c_void_p]),
    "gic_disc_fwd_redrop": (C.c_int, [_P(DiscDims), _P(DiscParams), _P(DiscShadow), _P(DiscState), _P(DiscState), C.c_int, c_void_p,
                            C.c_uint64, c_void_p, c_void_p, C.c_int, c_void_p]),
    "gic_disc_bwd": (C.c_int, [_P(DiscDims), _P(DiscParams), _P(DiscShadow), _P(DiscState), _P(DiscBwdWs), c_void_p,
                               C.c_int64, c_void_p, C.c_int, c_void_p, _P(DiscGrads), C.c_int, c_void_p, C.c_int64, c_void_p]),
    "gic_pack_image": (C.c_int, [c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "gic_repack_conv_weight": (C.c_int, [c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "gic_conv2d_bn_in": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_float, c_void_p, c_void_p, c_void_p, C.c_int,
                         C.c_int] + [C.c_int] * 9 + [c_void_p]),
    "gic_conv1x1_res_in": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_float,
                           c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "gic_conv1x1_bn_in_stats": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_float, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int64,
                                C.c_int, C.c_int, c_void_p]),
    "gic_conv_b2b": (C.c_int, [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                     C.c_int, c_void_p, c_void_p, C.c_float, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                     c_void_p]),
    "gic_conv2d": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int] + [C.c_int] * 9 + [c_void_p]),
    "gic_bn_act": (C.c_int, [c_void_p] * 12 + [C.c_int, C.c_float, C.c_int, c_void_p, C.c_int, C.c_int64, C.c_int, c_void_p]),
    "gic_bn_relu_maxpool": (C.c_int, [c_void_p] * 6 + [C.c_int, C.c_float, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "gic_avgpool": (C.c_int, [c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "gic_bn_running_update": (C.c_int, [c_void_p, C.c_int, c_void_p]),
    "gic_bn1d_fwd": (C.c_int, [c_void_p] * 5 + [C.c_int, C.c_float, C.c_float, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, c_void_p]),
    "gic_bn1d_bwd": (C.c_int, [c_void_p] * 4 + [C.c_int, c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, c_void_p]),
    "gic_colsum": (C.c_int, [c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, c_void_p, C.c_int, c_void_p]),
    "gic_gan_losses": (C.c_int, [C.c_int, c_void_p, c_void_p, c_void_p, C.c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "gic_xent": (C.c_int, [c_void_p, C.c_int, C.c_int64, C.c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gic_rollout_rewards": (C.c_int, [c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "gic_clip_adam_partials": (C.c_int64, [C.c_int64]),
    "gic_clip_adam": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
}

ABI_VERSION = 4               # GIC_ABI_VERSION of include/gicap.h
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


class GicError(RuntimeError):
    pass


def library_path() -> str:
    return _build.LIB


def load() -> C.CDLL:
    """Load libgicap.so; raise (never fall back) if it is missing or exports are incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise GicError(f"{path} not found: build it first (python __graft_entry__.py, or gan-image-captioning_amd/build.py); "
                       "there is no CPU / eager fallback for the hot path")
    lib = C.CDLL(path)
    for name, (res, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GicError(f"{path} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = argtypes
    if lib.gic_abi_version() != ABI_VERSION:
        raise GicError(f"ABI version mismatch: library {lib.gic_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


ERR_UNSUPPORTED = -2          # enum gic_status GIC_STATUS_UNSUPPORTED


def check(status: int, what: str = "") -> None:
    if status == 0:
        return
    msg = load().gic_last_error().decode(errors="replace")
    text = f"libgicap {what} failed with status {status}: {msg}"
    if status == -1:
        raise ValueError(text)
    if status == -2:
        raise NotImplementedError(text)
    raise GicError(text)
