"""ctypes binding of libdpsx.so (C ABI: include/dpsx.h).

The library is the product: there is no Python/torch fallback for any entry
point.  If the shared object is missing or a call fails, this module raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPSX_LIB: development aid to A/B an experimental build of the same library (tools/abl.sh)
SO_PATH = os.environ.get("DPSX_LIB") or os.path.join(_HERE, "lib", "libdpsx.so")

OK, EINVAL, EUNSUPPORTED, ELAUNCH, ENOMEM, EWORKSPACE = 0, -1, -2, -3, -4, -5
KIND_TAPS, KIND_SEP, KIND_RESIZE, KIND_MASK, KIND_IDENT, KIND_PHASE = range(6)
BLUR_AUTO, BLUR_FORCE_TAPS = 0, 1
POTENTIALS = {"mean": 1, "min": 2, "diff": 3, "curr": 4}      # DPSX_POT_*
ABI_VERSION = 3          # DPSX_ABI_VERSION of include/dpsx.h


class Coefs(ctypes.Structure):
    """struct dpsx_coefs"""
    _fields_ = [("a", c_float), ("b", c_float), ("c1", c_float), ("c2", c_float),
                ("min_log", c_float), ("max_log", c_float), ("add_noise", c_int32)]


class DpsxError(RuntimeError):
    def __init__(self, code, where):
        lib = _lib
        msg = lib.dpsx_strerror(code).decode() if lib is not None else str(code)
        hip = lib.dpsx_last_hip_error().decode() if lib is not None and code in (ELAUNCH, ENOMEM) else ""
        super().__init__(f"{where}: {msg} ({code})" + (f" [{hip}]" if hip else ""))
        self.code = code


_f = c_void_p      # device float*
_p = c_void_p      # any device pointer / stream
_i64 = c_int64

# name -> (restype, argtypes); mirrors include/dpsx.h one to one
SIGNATURES = {
    "dpsx_abi_version": (c_int, []),
    "dpsx_strerror": (c_char_p, [c_int]),
    "dpsx_last_hip_error": (c_char_p, []),
    "dpsx_posterior_fwd_f32": (c_int, [_f, _f, _f, _f, _f, _p, _i64, _i64, POINTER(Coefs), _p]),
    "dpsx_posterior_bwd_f32": (c_int, [_f, _f, _f, _f, _f, _f, _f, _i64, _i64, POINTER(Coefs), _p]),
    "dpsx_op_create_blur": (c_int, [POINTER(c_float), c_int, c_int, POINTER(c_void_p)]),
    "dpsx_op_create_resize": (c_int, [_i64, _i64, POINTER(c_float), POINTER(c_int64), _i64, _i64,
                                      POINTER(c_float), POINTER(c_int64), _i64, _i64, POINTER(c_void_p)]),
    "dpsx_op_create_mask": (c_int, [_f, _i64, _i64, POINTER(c_void_p)]),
    "dpsx_op_create_identity": (c_int, [POINTER(c_void_p)]),
    "dpsx_op_create_phase": (c_int, [_i64, _i64, _i64, POINTER(c_void_p)]),
    "dpsx_op_destroy": (None, [c_void_p]),
    "dpsx_op_out_shape": (c_int, [c_void_p, _i64, _i64, POINTER(c_int64), POINTER(c_int64)]),
    "dpsx_op_kind": (c_int, [c_void_p]),
    "dpsx_op_workspace_bytes": (_i64, [c_void_p, _i64, _i64, _i64, _i64]),
    "dpsx_op_forward_f32": (c_int, [c_void_p, _f, _f, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "dpsx_op_adjoint_f32": (c_int, [c_void_p, _f, _f, _f, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "dpsx_residual_norm_f32": (c_int, [_f, _i64, _f, _f, _f, _i64, _i64, _p, _i64, _p]),
    "dpsx_norm_bwd_f32": (c_int, [_f, _f, _f, c_int, _f, _i64, _i64, _p]),
    "dpsx_step_resid_bytes": (_i64, [c_void_p, _i64, _i64, _i64, _i64]),
    "dpsx_step_fwd_f32": (c_int, [c_void_p, _f, _f, _f, _f, _i64, _f, _f, _p, _p, _f,
                                  _i64, _i64, _i64, _i64, POINTER(Coefs), _p, _i64, _p]),
    "dpsx_step_bwd_f32": (c_int, [c_void_p, _p, _f, _f, _p, _f, _f, _i64, c_float, c_int, _f,
                                  _i64, _i64, _i64, _i64, POINTER(Coefs), _p, _i64, _p]),
    "dpsx_step_bwd_extra_f32": (c_int, [c_void_p, _p, _f, _f, _p, _f, _f, _i64, c_float, c_int, _f, _f,
                                        _i64, _i64, _i64, _i64, POINTER(Coefs), _p, _i64, _p]),
    "dpsx_step_update_f32": (c_int, [_f, _f, _f, _f, _i64, _i64, POINTER(Coefs), _p]),
    "dpsx_update_f32": (c_int, [_f, _f, _f, _f, _i64, _p]),
    "dpsx_score_f32": (c_int, [c_void_p, _f, _f, _i64, _f, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "dpsx_score_argmin_f32": (c_int, [c_void_p, _f, _f, _i64, _f, _p, _f, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "dpsx_search_step_f32": (c_int, [c_void_p, _f, _f, _f, _f, _i64, _f, _f, _p, _f, _f, _i64, _i64, _i64, _i64,
                                     POINTER(Coefs), _p, _i64, _p]),
    "dpsx_search_step_one_f32": (c_int, [c_void_p, _f, _f, _f, _f, _i64, _f, _f, _p, _f, _f, _i64, _i64, _i64, _i64,
                                         POINTER(Coefs), _p, _i64, _p]),
    "dpsx_resample_cost_f32": (c_int, [c_void_p, _f, _f, _i64, _f, c_int, _f, _f, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "dpsx_argmin_f32": (c_int, [_f, _i64, _p, _f, _p]),
    "dpsx_gather_f32": (c_int, [_f, _p, _f, _i64, _i64, _i64, _p]),
    "dpsx_replicate_f32": (c_int, [_f, _p, _f, _i64, _i64, _i64, _p]),
    "dpsx_pack_champion_f32": (c_int, [_f, _f, _p, _f, _f, _i64, _i64, _p]),
    "dpsx_select_champion_f32": (c_int, [_f, _i64, _i64, _f, _i64, _p, _p, _p]),
}

_lib = None


def lib():
    """Load libdpsx.so (built by __graft_entry__.build() / dps_ttc_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: the HIP extension is the only implementation of the DPS hot path "
                "(build it with `python -c 'import __graft_entry__ as g; g.build()'`)")
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)     # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if handle.dpsx_abi_version() != ABI_VERSION:
            raise RuntimeError("libdpsx ABI version mismatch")
        _lib = handle
    return _lib


def check(code, where):
    if code != OK:
        raise DpsxError(code, where)


def require_cuda(t, what="tensor"):
    if not t.is_cuda:
        raise RuntimeError(
            f"dps_ttc_amd: {what} is on {t.device}; the DPS hot path runs only as HIP kernels on an MI355X "
            "(no CPU fallback exists by design)")


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream_of(t):
    import torch
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def f32c(t, what="tensor"):
    """contiguous fp32 view/copy on the same device"""
    import torch
    require_cuda(t, what)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
