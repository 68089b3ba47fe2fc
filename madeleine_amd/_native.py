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
ABI_VERSION = 24   # == MDL_ABI_VERSION of include/madeleine_amd.h this file's SIGNATURES were written against

c_f = ctypes.c_void_p  # float* (device)
c_p = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int
u64 = ctypes.c_uint64
f32 = ctypes.c_float

# name -> (restype, argtypes); mirrors include/madeleine_amd.h declaration by declaration
SIGNATURES = {
    "mdl_version": (ctypes.c_char_p, []),
    "mdl_abi_version": (i32, []),
    "mdl_abmil_gate_fwd_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_gate_fwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_p, c_p]),
    "mdl_abmil_gate_bwd_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_gate_bwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32,
                                 u64, c_p, c_p, c_p, c_p]),
    "mdl_abmil_attnpool_bwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_f, c_f,
                                     c_f, c_f, c_p, i64, c_p, c_p]),
    "mdl_abmil_attnpool_bwd_phases": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p,
                                            c_f, c_f, c_f, c_f, c_p, i64, c_p, c_p, i32]),
    "mdl_abmil_attnpool_bwd_phases_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p,
                                                 c_p, c_f, c_f, c_f, c_f, c_p, i64, c_p, c_p, i32]),
    "mdl_abmil_gate_dropout_mask": (i32, [c_p, i64, i32, i32, f32, u64, c_p]),
    "mdl_abmil_pool_ws_bytes": (i64, [i64, i64, i32]),
    "mdl_abmil_pool_fwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_pool_bwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, i32, i64, i64, c_p, i64, i32, c_p]),
    "mdl_abmil_pool_view_fwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_pool_view_bwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p]),
    "mdl_abmil_pool_view_fwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_pool_view_bwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p]),
    "mdl_ln_gelu_drop_fwd": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, f32, u64, c_p, c_p]),
    "mdl_ln_gelu_drop_bwd_ws_bytes": (i64, [i64, i32]),
    "mdl_ln_gelu_drop_bwd": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_p]),
    "mdl_linear_fwd_ws_bytes": (i64, [i64, i32, i32]),
    "mdl_linear_fwd": (i32, [c_f, i64, c_f, c_f, c_f, i64, i64, i32, i32, c_p, c_p]),
    "mdl_linear_bwd_ws_bytes": (i64, [i64, i32, i32]),
    "mdl_linear_bwd": (i32, [c_f, i64, c_f, c_f, i64, c_f, i64, c_f, c_f, i64, i32, i32, c_p, c_p]),
    "mdl_infonce_ws_bytes": (i64, [i32, i32, i32]),
    "mdl_infonce_fwd": (i32, [c_f, c_f, c_p, c_f, c_f, i32, i32, i32, f32, i32, c_p, c_p]),
    "mdl_infonce_bwd": (i32, [c_f, c_f, c_f, c_f, c_p, c_f, c_f, i32, i32, i32, f32, i32, c_p, c_p]),
    "mdl_got_ws_bytes": (i64, [i32, i32, i32]),
    "mdl_got_fwd": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_extrema": (i32, [c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_bwd": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_bwd_begin": (i32, [c_f, c_f, i32, i32, i32, c_p, c_p]),
    "mdl_got_bwd_finish": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_p, c_p]),
    # several problems per launch: host arrays of np device pointers / ints
    "mdl_got_extrema_multi": (i32, [i32, c_p, c_p, c_p, c_p, c_p, i32, c_p, c_p]),
    "mdl_got_fwd_multi": (i32, [i32, c_p, c_p, c_p, c_p, c_p, c_p, i32, c_p, c_p]),
    "mdl_got_bwd_begin_multi": (i32, [i32, c_p, c_p, c_p, c_p, i32, c_p, c_p]),
    "mdl_got_bwd_finish_multi": (i32, [i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i32, c_p, c_p]),
    # bf16 mode (same argument lists as the fp32 entry points; activation pointers are bf16)
    "mdl_ln_gelu_drop_fwd_bf16": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, f32, u64, c_p, c_p]),
    "mdl_ln_gelu_drop_bwd_bf16": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p,
                                        c_p]),
    "mdl_ln_gelu_drop_fwd_groups": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, f32, u64, c_p, c_p, i32, c_p]),
    "mdl_ln_gelu_drop_fwd_groups_bf16": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, f32, u64, c_p, c_p, i32, c_p]),
    "mdl_ln_gelu_drop_bwd_groups": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, i32,
                                          c_p, c_p]),
    "mdl_ln_gelu_drop_bwd_groups_bf16": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, i32,
                                               c_p, c_p]),
    "mdl_abmil_pool_fwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_pool_bwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, i32, i64, i64, c_p, i64, i32, c_p]),
    "mdl_abmil_wpool_fwd": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_wpool_bwd": (i32, [c_f, i64, c_f, c_f, c_f, i32, c_f, i64, i64, c_p, i64, i32, c_p]),
    "mdl_abmil_wpool_fwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_wpool_bwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, i32, c_f, i64, i64, c_p, i64, i32, c_p]),
    "mdl_linear_bf16_supported": (i32, [i64, i64, i32]),
    "mdl_linear_fwd_bf16_ws_bytes": (i64, [i64, i64, i64]),
    "mdl_linear_fwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, i64, i64, i64, i64, c_p, c_p]),
    "mdl_linear_bwd_bf16_ws_bytes": (i64, [i64, i64, i64]),
    "mdl_linear_bwd_bf16": (i32, [c_f, i64, c_f, c_f, i64, c_f, i64, c_f, c_f, i64, i64, i64, c_p, c_p]),
    "mdl_abmil_gate_fwd_bf16_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_gate_fwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_p,
                                      c_p]),
    "mdl_abmil_gate_bwd_bf16_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_attnpool_bwd_bf16": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_f, c_f,
                                     c_f, c_f, c_p, i64, c_p, c_p]),
    # split-fp16 engine
    "mdl_pool_timer_arm": (i32, [i32]),
    "mdl_pool_timer_read": (i32, [i32, c_f]),
    "mdl_stream_create_cu_mask": (i32, [ctypes.c_uint32, c_p, c_p]),
    "mdl_stream_destroy": (i32, [c_p]),
    "mdl_split_image": (i32, [c_f, i64, i64, i32, c_p, i64, i64, c_f, c_p]),
    "mdl_abmil_pool_fwd_img": (i32, [c_p, i64, c_f, c_f, c_f, c_f, c_f, i64, i64, c_p, i64, i32, c_p, c_p]),
    "mdl_abmil_pool_dscores_img": (i32, [c_p, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i64, i64, c_p, i64, i32, c_p]),
    "mdl_split_tile_absmax": (i32, [c_f, i64, i64, i32, c_f, c_f, c_p]),
    "mdl_split_gemm_nt": (i32, [c_p, i64, c_f, c_p, i64, c_f, c_f, i64, i64, i32, i32, c_f, i32, c_f, c_f, c_f, c_f, i32, c_p]),
    "mdl_split_gemm_nt_group_bias": (i32, [c_p, i64, c_f, c_p, i64, c_f, c_f, i64, i64, i32, i32, c_f, c_f, c_f, c_f, c_p, i32, c_p]),
    "mdl_ln_gelu_drop_bwd_groups_ws_bytes": (i64, [i64, i32, i32]),
    "mdl_ln_gelu_drop_bwd_split_groups": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_f, c_f,
                                                c_p, i32, c_f, c_p, c_p]),
    "mdl_split_image_rows": (i32, [c_f, i64, i64, i32, c_p, i64, i64, c_f, c_f, c_p]),
    "mdl_split_gemm_tn_ws_bytes": (i64, [i64, i32, i32]),
    "mdl_split_gemm_tn": (i32, [c_p, i64, c_f, i32, c_p, i64, c_f, i32, c_f, i64, c_f, c_p, i32, c_p]),
    "mdl_ln_gelu_drop_fwd_split": (i32, [c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_f, c_f, i64, i32, f32, f32, u64, c_p, c_f, c_f, c_p]),
    "mdl_ln_gelu_drop_bwd_split": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_f, c_f, c_p, c_p]),
    "mdl_abmil_gate_fwd_split_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_gate_fwd_split": (i32, [c_p, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, f32, u64, c_p, c_p, c_p, c_p]),
    "mdl_abmil_gate_bwd_split_ws_bytes": (i64, [i64, i32]),
    "mdl_abmil_attnpool_bwd_split": (i32, [c_p, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32,
                                           f32, u64, c_p, c_p, c_f, c_f, c_f, c_f, c_p, i64, c_f, c_p, c_p, i32, i32]),
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
        override = os.environ.get("MADELEINE_LIB")   # a library built elsewhere (A/B of two builds); the ABI check below still applies
        path = override or lib_path()
        if not override and not _build.is_fresh():
            try:
                _build.build()
            except _build.HipccMissing as e:
                # no compiler on this machine: a prebuilt library may still be used, guarded by the ABI check below
                if not os.path.exists(path):
                    raise RuntimeError(
                        "madeleine_amd: libmadeleine_amd.so is missing and could not be built (%s). "
                        "There is no fallback path: run `python -m madeleine_amd._build`." % e) from e
            except Exception as e:
                # sources newer than the library and the rebuild FAILED: never dlopen the stale object -- its entry points
                # would be called with this file's (newer) signatures
                raise RuntimeError("madeleine_amd: libmadeleine_amd.so is stale and the rebuild failed: %s" % e) from e
        try:
            handle = ctypes.CDLL(path)
        except OSError as e:
            raise RuntimeError("madeleine_amd: cannot load %s: %s" % (path, e)) from e
        try:
            abi = handle.mdl_abi_version
        except AttributeError:
            raise RuntimeError("madeleine_amd: %s predates the ABI-version check; rebuild it "
                               "(`python -m madeleine_amd._build --force`)" % path) from None
        abi.restype, abi.argtypes = i32, []
        if abi() != ABI_VERSION:
            raise RuntimeError("madeleine_amd: %s implements ABI revision %d, this binding expects %d; rebuild it "
                               "(`python -m madeleine_amd._build --force`)" % (path, abi(), ABI_VERSION))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise RuntimeError("madeleine_amd: %s does not export %s (header / library mismatch)" % (path, name)) from None
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
