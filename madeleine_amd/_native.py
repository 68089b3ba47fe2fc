"""ctypes binding of libmadeleine_amd.so (the C ABI of include/madeleine_amd.h).

Single backend: there is no CPU / eager fallback.  `lib()` raises if the shared object cannot be
loaded (it is built in-tree by `python -m madeleine_amd._build` / __graft_entry__.build(); if it is
missing or stale and hipcc is present it is rebuilt once), and every op raises RuntimeError on a
non-zero return code.
"""
import ctypes
import os
import threading

from . import _build

_LOCK = threading.Lock()
_LIB = None

c_f = ctypes.c_void_p  # float* (device)
c_p = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int
u64 = ctypes.c_uint64
f32 = ctypes.c_float

# name -> (restype, argtypes); mirrors include/madeleine_amd.h declaration by declaration
SIGNATURES = {
    "mdl_version": (ctypes.c_char_p, []),
    "mdl_abmil_gate_fwd_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_gate_fwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_p, c_p]),
    "mdl_abmil_gate_bwd_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_gate_bwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32,
                                 u64, c_p, c_p, c_p, c_p]),
    "mdl_abmil_attnpool_bwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_f, c_f,
                                     c_f, c_f, c_p, i64, c_p, c_p]),
    "mdl_abmil_gate_dropout_mask": (i32, [c_p, i64, i32, i32, f32, u64, c_p]),
    "mdl_abmil_pool_ws_bytes": (i64, [i64, i64, i32]),
    "mdl_abmil_pool_fwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_pool_bwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, i32, i64, i64, c_p, i64, i32, c_p]),
    "mdl_ln_gelu_drop_fwd": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, f32, u64, c_p, c_p]),
    "mdl_ln_gelu_drop_bwd_ws_bytes": (i64, [i64, i32]),
    "mdl_ln_gelu_drop_bwd": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_p]),
    "mdl_linear_fwd_ws_bytes": (i64, [i64, i32, i32]),
    "mdl_linear_fwd": (i32, [c_f, i64, c_f, c_f, i64, i64, i32, i32, c_p, c_p]),
    "mdl_linear_bwd_ws_bytes": (i64, [i64, i32, i32]),
    "mdl_linear_bwd": (i32, [c_f, i64, c_f, c_f, i64, c_f, i64, c_f, i64, i32, i32, c_p, c_p]),
    "mdl_infonce_ws_bytes": (i64, [i32, i32, i32]),
    "mdl_infonce_fwd": (i32, [c_f, c_f, c_p, c_f, i32, i32, i32, f32, i32, c_p, c_p]),
    "mdl_infonce_bwd": (i32, [c_f, c_p, c_f, c_f, i32, i32, i32, f32, i32, c_p, c_p]),
    "mdl_got_ws_bytes": (i64, [i32, i32, i32]),
    "mdl_got_fwd": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_extrema": (i32, [c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_bwd": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_bwd_begin": (i32, [c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_bwd_finish": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    # bf16 mode (same argument lists as the fp32 entry points; activation pointers are bf16)
    "mdl_ln_gelu_drop_fwd_bf16": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, f32, u64, c_p, c_p]),
    "mdl_ln_gelu_drop_bwd_bf16": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p,
                                        c_p]),
    "mdl_abmil_pool_fwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_pool_bwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, i32, i64, i64, c_p, i64, i32, c_p]),
    "mdl_abmil_gate_fwd_bf16_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_gate_fwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_p,
                                      c_p]),
    "mdl_abmil_gate_bwd_bf16_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_attnpool_bwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_f, c_f,
                                     c_f, c_f, c_p, i64, c_p, c_p]),
    "mdl_abmil_gate_bwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32,
                                      f32, u64, c_p, c_p, c_p, c_p]),
}


def lib_path() -> str:
    return _build.LIB


def lib():
    """Load (once) and return the ctypes handle.  Raises RuntimeError if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = lib_path()
        if not _build.is_fresh():
            try:
                _build.build()
            except Exception as e:  # no hipcc, or compile error
                if not os.path.exists(path):
                    raise RuntimeError(
                        "madeleine_amd: libmadeleine_amd.so is missing and could not be built (%s). "
                        "There is no fallback path: run `python -m madeleine_amd._build`." % e) from e
        try:
            handle = ctypes.CDLL(path)
        except OSError as e:
            raise RuntimeError("madeleine_amd: cannot load %s: %s" % (path, e)) from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
        return _LIB


_ERR = {-1: "MDL_E_ARG (bad size / null pointer)", -2: "MDL_E_ALIGN (pointer not 16-byte aligned)",
        -3: "MDL_E_UNSUPPORTED"}


def check(rc: int, what: str):
    if rc != 0:
        msg = _ERR.get(rc, "hipError_t %d" % rc)
        raise RuntimeError("madeleine_amd: %s failed: %s" % (what, msg))
