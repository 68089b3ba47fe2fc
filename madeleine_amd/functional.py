"""torch.autograd.Function wrappers over the C ABI (include/madeleine_amd.h).

PyTorch is plumbing here: it owns device memory, the stream and the autograd graph; every numeric
step of the hot path below runs in libmadeleine_amd.so.  Tensors are contiguous, on a ROCm device, fp32 --
or, for the activation tensors of the bf16 mode (E, LayerNorm input/output), bfloat16, which selects the *_bf16 entry
points; anything else raises (there is no CPU or eager fallback).

Layouts (see include/madeleine_amd.h): token embeddings are head-major [T, H*512]; scores [T, H].
"""
import os
import threading
from typing import Optional

import torch

from . import _native

HID = 512


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


ACT_DTYPES = (torch.float32, torch.bfloat16)   # storage types of the activation tensors (parity mode / bf16 mode)


def _sfx(t: torch.Tensor) -> str:
    """C-ABI entry-point suffix for the storage type of an activation tensor."""
    return "_bf16" if t.dtype == torch.bfloat16 else ""


def _require_act(t: torch.Tensor, name: str):
    if t.dtype not in ACT_DTYPES:
        raise RuntimeError("madeleine_amd: %s must be float32 or bfloat16 (got %s)" % (name, t.dtype))
    return _require(t, name, t.dtype)


def _require(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(
            "madeleine_amd: %s must live on a ROCm device (got %s); the HIP kernels are the only backend" % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("madeleine_amd: %s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("madeleine_amd: %s must be contiguous" % name)
    return t


class KernelTimer:
    """Optional per-call HIP-event timing of the C-ABI launches (bench.py's roofline leg).  Events are recorded
    on the stream the kernels are launched on (torch's current stream); nothing is synchronised until report().
    A call may declare its algorithmic work -- ("flop", n) for the matrix-core contractions, ("byte", n) for the HBM-bound
    passes -- so that bench.py can print achieved TFLOP/s / GB/s per kernel family (work())."""

    def __init__(self, only=None):
        """only: names to time (None = every call).  bench.py's headline region times the A3 forward alone -- one event pair per
        step -- so that the fused backward runs as ONE call and no per-launch events sit in the measured step."""
        self.pairs = {}
        self.works = {}
        self.only = None if only is None else frozenset(only)

    def wants(self, name) -> bool:
        return self.only is None or name in self.only

    def start(self, name, work=None):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.pairs.setdefault(name, []).append((ev0, ev1))
        if work is not None:
            kind, amount = work
            w = self.works.setdefault(name, [kind, 0.0, 0])
            w[1] += float(amount)
            w[2] += 1
        ev0.record()
        return ev1

    def report(self):
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in self.pairs.items()}

    def work(self):
        """{name: (kind, work per call on average)} for the calls that declared it."""
        return {k: (w[0], w[1] / max(1, w[2])) for k, w in self.works.items()}


TIMER: Optional[KernelTimer] = None


class PoolDispatchTimer:
    """Execution time of the A3 forward's two kernels (pool_partial, pool_combine) from the dispatches' own begin / end events
    (mdl_pool_timer_arm / _read: hipExtLaunchKernel start / stop events on the launch stream) -- what rocprofv3 --kernel-trace
    reports for them, not the distance between two markers in a busy stream.  Every pooling forward issued while POOL_TIMER is set
    takes the next of 64 slots; nothing is synchronised until report()."""
    SLOTS = 64

    def __init__(self):
        self.n = 0

    def arm(self):
        _native.check(_native.lib().mdl_pool_timer_arm(self.n % self.SLOTS), "mdl_pool_timer_arm")
        self.n += 1

    def report(self):
        """[(pool_partial ms, pool_combine ms, first start -> last end ms)] of the last min(calls, 64) pooling forwards."""
        import ctypes
        lib, out = _native.lib(), []
        for i in range(max(0, self.n - self.SLOTS), self.n):
            ms = (ctypes.c_float * 3)()
            _native.check(lib.mdl_pool_timer_read(i % self.SLOTS, ms), "mdl_pool_timer_read")
            out.append((float(ms[0]), float(ms[1]), float(ms[2])))
        return out


POOL_TIMER: Optional[PoolDispatchTimer] = None


class _timed:
    def __init__(self, name, work=None):
        self.name, self.work = name, work

    def __enter__(self):
        self.ev = TIMER.start(self.name, self.work) if TIMER is not None and TIMER.wants(self.name) else None
        if POOL_TIMER is not None and self.name == "pool_fwd":
            POOL_TIMER.arm()

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev.record()
        return False


def _ws(nbytes: int, device) -> torch.Tensor:
    if nbytes < 0:
        raise RuntimeError("madeleine_amd: workspace query failed (%d)" % nbytes)
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def h2d(t: torch.Tensor, device, dtype=None) -> torch.Tensor:
    """Host tensor -> device WITHOUT blocking the host.  A copy from pageable memory (`t.to(device, non_blocking=True)` on an
    ordinary CPU tensor) is stream-ordered AND synchronous for the caller: the host stalls until the GPU has drained everything queued
    before it -- in the pretrain step that was the whole encoder forward (8 ms at config 2, 21 ms at config 3: the host then fed
    the loss section launch by launch and the device idled between them).  Staged through a pinned buffer (PyTorch's caching host
    allocator keeps it alive until the copy has run) the copy is a queued DMA and the host stays ahead of the device."""
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    dev = torch.device(device)
    if t.device.type != "cpu" or dev.type != "cuda" or os.environ.get("MADELEINE_BLOCKING_H2D"):
        return t.to(dev)
    return t.contiguous().pin_memory().to(dev, non_blocking=True)


def new_dropout_seed() -> int:
    """Draws a 63-bit seed from torch's CPU generator (deterministic under torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


# --------------------------------------------------------------------------------------------------
# raw calls (no autograd)
# --------------------------------------------------------------------------------------------------
def gate_fwd_raw(E2d, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, save_act: bool):
    lib = _native.lib()
    T, H = E2d.shape[0], Wa.shape[0]
    dev = E2d.device
    scores = torch.empty(T, H, device=dev, dtype=torch.float32)
    act_a = torch.empty(T, H, HID, device=dev, dtype=E2d.dtype) if save_act else None
    act_b = torch.empty(T, H, HID, device=dev, dtype=E2d.dtype) if save_act else None
    sfx = _sfx(E2d)
    ws = _ws(getattr(lib, "mdl_abmil_gate_fwd%s_ws_bytes" % sfx)(T, H), dev)
    with _timed("gate_fwd", ("flop", 2.0 * T * H * HID * 2 * HID)):
        rc = getattr(lib, "mdl_abmil_gate_fwd" + sfx)(_ptr(E2d), E2d.stride(0), _ptr(Wa), _ptr(ba), _ptr(Wb), _ptr(bb), _ptr(wc), _ptr(bc),
                                    _ptr(scores), _ptr(act_a), _ptr(act_b), T, H, float(p_drop), int(seed),
                                    _ptr(keep_a), _ptr(keep_b), _ptr(ws), _stream())
    _native.check(rc, "mdl_abmil_gate_fwd")
    return scores, act_a, act_b


def gate_fwd_split_raw(Ei, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, save_act: bool):
    """gate_fwd_raw on the split engine: Ei = SplitImage of the head-major token embeddings [T, H*512]."""
    lib = _native.lib()
    T, H = Ei.rows, Wa.shape[0]
    dev = Ei.data.device
    scores = torch.empty(T, H, device=dev, dtype=torch.float32)
    act_a = torch.empty(T, H, HID, device=dev, dtype=torch.float32) if save_act else None
    act_b = torch.empty(T, H, HID, device=dev, dtype=torch.float32) if save_act else None
    ws = _ws(lib.mdl_abmil_gate_fwd_split_ws_bytes(T, H), dev)
    with _timed("gate_fwd", ("flop", 2.0 * T * H * HID * 2 * HID)):
        rc = lib.mdl_abmil_gate_fwd_split(_ptr(Ei.data), Ei.K * 4, _ptr(Ei.scale), _ptr(Wa), _ptr(ba), _ptr(Wb), _ptr(bb), _ptr(wc), _ptr(bc),
                                          _ptr(scores), _ptr(act_a), _ptr(act_b), T, H, float(p_drop), int(seed), _ptr(keep_a), _ptr(keep_b),
                                          _ptr(ws), _stream())
    _native.check(rc, "mdl_abmil_gate_fwd_split")
    return scores, act_a, act_b


def attnpool_bwd_split_raw(Ei, Wa, Wb, wc, act_a, act_b, d_scores, dE, p_drop, seed, keep_a, keep_b, scores, stat_m, stat_l, d_pooled,
                           row_bag, N, accumulate=0, dE_absmax=None, phases=None):
    """attnpool_bwd_raw (scores None: gate_bwd_raw) on the split engine.  `phases`: a sequence of phase masks issued one after the other
    on the same workspace (include/madeleine_amd.h: 1 = dz pass, 2 = both contractions, 4 = dX alone, 8 = dW alone); None = one call."""
    lib = _native.lib()
    T, H = Ei.rows, Wa.shape[0]
    dev = Ei.data.device
    dWa, dWb = torch.empty_like(Wa), torch.empty_like(Wb)
    dba = torch.empty(H, HID, device=dev, dtype=torch.float32)
    dbb, dwc = torch.empty_like(dba), torch.empty_like(dba)
    dbc = torch.empty(H, device=dev, dtype=torch.float32)
    ws = _ws(lib.mdl_abmil_gate_bwd_split_ws_bytes(T, H), dev)

    def call(phases):
        rc = lib.mdl_abmil_attnpool_bwd_split(_ptr(Ei.data), Ei.K * 4, _ptr(Ei.scale), _ptr(Wa), _ptr(Wb), _ptr(wc), _ptr(act_a), _ptr(act_b),
                                              _ptr(d_scores), _ptr(dE), dE.stride(0), int(accumulate), _ptr(dWa), _ptr(dWb), _ptr(dba), _ptr(dbb),
                                              _ptr(dwc), _ptr(dbc), T, H, float(p_drop), int(seed), _ptr(keep_a), _ptr(keep_b), _ptr(scores),
                                              _ptr(stat_m), _ptr(stat_l), _ptr(d_pooled), _ptr(row_bag), int(N), _ptr(dE_absmax), _ptr(ws),
                                              _stream(), phases, GRAD_TERMS)
        _native.check(rc, "mdl_abmil_attnpool_bwd_split")
    if phases is not None:
        for ph in phases:
            call(int(ph))
    elif TIMER is not None and TIMER.wants("gate_bwd_dz"):
        with _timed("gate_bwd_dz", ("byte", float(T) * H * 4 * HID * 4)):
            call(1)
        with _timed("gate_bwd_gemm", ("flop", 4.0 * T * H * HID * 2 * HID)):
            call(2)
    else:
        call(3)
    return dWa, dWb, dba, dbb, dwc, dbc


def gate_bwd_raw(E2d, Wa, Wb, wc, act_a, act_b, d_scores, dE, accumulate, p_drop, seed, keep_a, keep_b):
    lib = _native.lib()
    T, H = E2d.shape[0], Wa.shape[0]
    dev = E2d.device
    dWa, dWb = torch.empty_like(Wa), torch.empty_like(Wb)
    dba = torch.empty(H, HID, device=dev, dtype=torch.float32)
    dbb, dwc = torch.empty_like(dba), torch.empty_like(dba)
    dbc = torch.empty(H, device=dev, dtype=torch.float32)
    sfx = _sfx(E2d)
    ws = _ws(getattr(lib, "mdl_abmil_gate_bwd%s_ws_bytes" % sfx)(T, H), dev)
    with _timed("gate_bwd", ("flop", 4.0 * T * H * HID * 2 * HID)):
        rc = getattr(lib, "mdl_abmil_gate_bwd" + sfx)(_ptr(E2d), E2d.stride(0), _ptr(Wa), _ptr(Wb), _ptr(wc), _ptr(act_a), _ptr(act_b),
                                    _ptr(d_scores), _ptr(dE), int(accumulate), _ptr(dWa), _ptr(dWb), _ptr(dba), _ptr(dbb),
                                    _ptr(dwc), _ptr(dbc), T, H, float(p_drop), int(seed), _ptr(keep_a), _ptr(keep_b),
                                    _ptr(ws), _stream())
    _native.check(rc, "mdl_abmil_gate_bwd")
    return dWa, dWb, dba, dbb, dwc, dbc


def attnpool_bwd_raw(E2d, Wa, Wb, wc, act_a, act_b, d_scores, dE, p_drop, seed, keep_a, keep_b, scores, stat_m, stat_l, d_pooled,
                     row_bag, N, accumulate=0):
    """Gate backward whose dX epilogue also adds the pooling term (mdl_abmil_attnpool_bwd): dE is written once; with
    `accumulate` the epilogue adds to what dE already holds (another consumer's gradient of E)."""
    lib = _native.lib()
    T, H = E2d.shape[0], Wa.shape[0]
    dev = E2d.device
    dWa, dWb = torch.empty_like(Wa), torch.empty_like(Wb)
    dba = torch.empty(H, HID, device=dev, dtype=torch.float32)
    dbb, dwc = torch.empty_like(dba), torch.empty_like(dba)
    dbc = torch.empty(H, device=dev, dtype=torch.float32)
    sfx = _sfx(E2d)
    ws = _ws(getattr(lib, "mdl_abmil_gate_bwd%s_ws_bytes" % sfx)(T, H), dev)
    args = (_ptr(E2d), E2d.stride(0), _ptr(Wa), _ptr(Wb), _ptr(wc), _ptr(act_a), _ptr(act_b), _ptr(d_scores), _ptr(dE), int(accumulate), _ptr(dWa),
            _ptr(dWb), _ptr(dba), _ptr(dbb), _ptr(dwc), _ptr(dbc), T, H, float(p_drop), int(seed), _ptr(keep_a), _ptr(keep_b),
            _ptr(scores), _ptr(stat_m), _ptr(stat_l), _ptr(d_pooled), _ptr(row_bag), int(N), _ptr(ws), _stream())
    if TIMER is not None and TIMER.wants("gate_bwd_dz"):
        # profiling: the HBM-bound dz pass and the MFMA-bound contractions as two calls on the same workspace, timed separately
        # ("gate_bwd" stays their sum in KernelTimer.report)
        fn = getattr(lib, "mdl_abmil_attnpool_bwd_phases" + sfx)
        with _timed("gate_bwd_dz", ("byte", float(T) * H * 4 * HID * E2d.element_size())):   # reads a, b; writes dza | dzb
            rc = fn(*args, 1)
        _native.check(rc, "mdl_abmil_attnpool_bwd_phases")
        with _timed("gate_bwd_gemm", ("flop", 4.0 * T * H * HID * 2 * HID)):
            rc = fn(*args, 2)
        _native.check(rc, "mdl_abmil_attnpool_bwd_phases")
    else:
        rc = getattr(lib, "mdl_abmil_attnpool_bwd" + sfx)(*args)
        _native.check(rc, "mdl_abmil_attnpool_bwd")
    return dWa, dWb, dba, dbb, dwc, dbc


def pool_fwd_raw(E2d, scores, n_bags, N, cu_seqlens, max_len):
    lib = _native.lib()
    H = scores.shape[-1]
    dev = E2d.device
    pooled = torch.empty(n_bags, H * HID, device=dev, dtype=torch.float32)
    stat_m = torch.empty(n_bags, H, device=dev, dtype=torch.float32)
    stat_l = torch.empty(n_bags, H, device=dev, dtype=torch.float32)
    ws = _ws(lib.mdl_abmil_pool_ws_bytes(n_bags, max_len, H), dev)
    T = E2d.shape[0]
    with _timed("pool_fwd", ("byte", float(T) * H * (HID * E2d.element_size() + 4) + n_bags * H * HID * 4.0)):
        rc = getattr(lib, "mdl_abmil_pool_fwd" + _sfx(E2d))(_ptr(E2d), E2d.stride(0), _ptr(scores), _ptr(pooled), _ptr(stat_m), _ptr(stat_l), n_bags,
                                    N, _ptr(cu_seqlens), max_len, H, _ptr(ws), _stream())
    _native.check(rc, "mdl_abmil_pool_fwd")
    return pooled, stat_m, stat_l


def pool_bwd_raw(E2d, scores, pooled, stat_m, stat_l, d_pooled, dE, accumulate, d_scores, accumulate_scores, n_bags, N,
                 cu_seqlens, max_len):
    lib = _native.lib()
    H = scores.shape[-1]
    T = E2d.shape[0]
    nb = float(T) * H * (HID * E2d.element_size() * (1 if dE is None else 2) + 8) + n_bags * H * HID * 4.0
    with _timed("pool_bwd", ("byte", nb)):
        rc = getattr(lib, "mdl_abmil_pool_bwd" + _sfx(E2d))(_ptr(E2d), E2d.stride(0), _ptr(scores), _ptr(pooled), _ptr(stat_m), _ptr(stat_l),
                                    _ptr(d_pooled), _ptr(dE), int(accumulate), _ptr(d_scores), int(accumulate_scores), n_bags, N,
                                    _ptr(cu_seqlens), max_len, H, _stream())
    _native.check(rc, "mdl_abmil_pool_bwd")


def pool_fwd_img_raw(Ei, scores, n_bags, N, cu_seqlens, max_len):
    """pool_fwd_raw on the split image of E (the split GEMM mode stores E as an image only)."""
    lib = _native.lib()
    H = scores.shape[-1]
    dev = scores.device
    pooled = torch.empty(n_bags, H * HID, device=dev, dtype=torch.float32)
    stat_m = torch.empty(n_bags, H, device=dev, dtype=torch.float32)
    stat_l = torch.empty(n_bags, H, device=dev, dtype=torch.float32)
    ws = _ws(lib.mdl_abmil_pool_ws_bytes(n_bags, max_len, H), dev)
    with _timed("pool_fwd", ("byte", float(Ei.rows) * H * (HID * 4 + 4) + n_bags * H * HID * 4.0)):
        rc = lib.mdl_abmil_pool_fwd_img(_ptr(Ei.data), Ei.K * 4, _ptr(Ei.scale), _ptr(scores), _ptr(pooled), _ptr(stat_m), _ptr(stat_l),
                                        n_bags, N, _ptr(cu_seqlens), max_len, H, _ptr(ws), _stream())
    _native.check(rc, "mdl_abmil_pool_fwd_img")
    return pooled, stat_m, stat_l


def pool_dscores_img_raw(Ei, scores, pooled, stat_m, stat_l, d_pooled, d_scores, accumulate_scores, n_bags, N, cu_seqlens, max_len):
    """The score gradients of the pooling from the split image of E (pool_bwd_raw with dE = None)."""
    lib = _native.lib()
    H = scores.shape[-1]
    with _timed("pool_bwd", ("byte", float(Ei.rows) * H * (HID * 4 + 8) + n_bags * H * HID * 4.0)):
        rc = lib.mdl_abmil_pool_dscores_img(_ptr(Ei.data), Ei.K * 4, _ptr(Ei.scale), _ptr(scores), _ptr(pooled), _ptr(stat_m), _ptr(stat_l),
                                            _ptr(d_pooled), _ptr(d_scores), int(accumulate_scores), n_bags, N, _ptr(cu_seqlens), max_len,
                                            H, _stream())
    _native.check(rc, "mdl_abmil_pool_dscores_img")


def pool_view_fwd_raw(E2d, scores, n_bags, N, token_idx):
    lib = _native.lib()
    H = scores.shape[-1]
    dev = E2d.device
    n_idx = token_idx.numel()
    pooled = torch.empty(n_bags, H * HID, device=dev, dtype=torch.float32)
    stat_m = torch.empty(n_bags, H, device=dev, dtype=torch.float32)
    stat_l = torch.empty(n_bags, H, device=dev, dtype=torch.float32)
    ws = _ws(lib.mdl_abmil_pool_ws_bytes(n_bags, n_idx, H), dev)
    with _timed("pool_view_fwd"):
        rc = getattr(lib, "mdl_abmil_pool_view_fwd" + _sfx(E2d))(_ptr(E2d), E2d.stride(0), _ptr(scores), _ptr(pooled), _ptr(stat_m),
                                                                _ptr(stat_l), n_bags, N, _ptr(token_idx), n_idx, H, _ptr(ws), _stream())
    _native.check(rc, "mdl_abmil_pool_view_fwd")
    return pooled, stat_m, stat_l


def pool_view_bwd_raw(E2d, scores, pooled, stat_m, stat_l, d_pooled, dE, d_scores, n_bags, N, token_idx):
    """Accumulates the view's contribution into dE and / or d_scores (either may be None)."""
    lib = _native.lib()
    H = scores.shape[-1]
    with _timed("pool_view_bwd"):
        rc = getattr(lib, "mdl_abmil_pool_view_bwd" + _sfx(E2d))(_ptr(E2d), E2d.stride(0), _ptr(scores), _ptr(pooled), _ptr(stat_m),
                                                                _ptr(stat_l), _ptr(d_pooled), _ptr(dE), _ptr(d_scores), n_bags, N,
                                                                _ptr(token_idx), token_idx.numel(), H, _stream())
    _native.check(rc, "mdl_abmil_pool_view_bwd")


# --------------------------------------------------------------------------------------------------
# split-fp16 engine (include/madeleine_amd.h, csrc/split_engine.hpp): fp32-accurate contractions on v_mfma_f32_32x32x16_f16
# --------------------------------------------------------------------------------------------------
GEMM_MODE = os.environ.get("MADELEINE_GEMM", "split")   # "split": 3-term split-fp16 products; "fp32": v_mfma_f32_32x32x2_f32


def gemm_mode() -> str:
    return GEMM_MODE


# Activation recomputation for the fused A2 + A3 node (opt-in; MADELEINE_GATE_RECOMPUTE=1 or set_gate_recompute(True)): the gate forward
# then saves NO tanh / sigmoid activations (2 x 8 KiB per token at H = 4: 4.3 GB per config-2 step, 10.7 GB at config 3) and the
# backward runs the gate forward once more (same seed -> same dropout masks, same bits) to rebuild them -- memory for time
# (+1 gate forward per step).  The default keeps the activations: with 288 GB of HBM the step's 60 GiB at config 3 fit many times over.
GATE_RECOMPUTE = os.environ.get("MADELEINE_GATE_RECOMPUTE", "0") == "1"


# Matrix terms of the split engine's BACKWARD products (opt-in; MADELEINE_GRAD_TERMS=2 or set_gradient_terms(2)).  3: every product is
# ah bh + ah bl + al bh.  2: in dX = dY W the weight, and in dW = dY^T X the activations, enter rounded to their hi plane (11 bits) --
# a third fewer matrix instructions in the backward, gradients at ~2^-12 relative instead of ~2^-22; the forward (every value the
# losses and the caller see) is untouched.  Measured in DESIGN.md 3.7; never the default.
GRAD_TERMS = int(os.environ.get("MADELEINE_GRAD_TERMS", "3"))


def set_gradient_terms(terms: int):
    global GRAD_TERMS
    if terms not in (2, 3):
        raise ValueError("gradient terms: 2 or 3")
    GRAD_TERMS = int(terms)


def set_gate_recompute(on: bool):
    global GATE_RECOMPUTE
    GATE_RECOMPUTE = bool(on)


def set_gemm_mode(mode: str):
    """'split' (default): the fp32 contractions run as ah bh + ah bl + al bh on the fp16 matrix cores (fp32-level accuracy, ~2x the
    rate); 'fp32': the exact-fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32)."""
    global GEMM_MODE
    if mode not in ("split", "fp32"):
        raise ValueError("gemm mode must be 'split' or 'fp32'")
    GEMM_MODE = mode


class SplitImage:
    """Split image of a [rows, K] fp32 tensor: `data` float32 [rows + pad, K] (opaque bytes: per 32-column block the fp16 hi plane
    | lo plane), `scale` float32 [2] = {scale, absmax} on the device."""
    __slots__ = ("data", "scale", "rows", "K", "row_inv")

    def __init__(self, data, scale, rows, K, row_inv=None):
        # row_inv [rows] (row-scaled images only): 1 / the power-of-two scale of every row; the common `scale` of such an image is 1
        self.data, self.scale, self.rows, self.K, self.row_inv = data, scale, rows, K, row_inv


def split_image(x2d, pad_rows=0) -> SplitImage:
    _require(x2d, "x")
    lib = _native.lib()
    rows, K = x2d.shape
    data = torch.empty(rows + pad_rows, K, device=x2d.device, dtype=torch.float32)
    scale = torch.empty(2, device=x2d.device, dtype=torch.float32)
    with _timed("split_image", ("byte", 12.0 * rows * K)):   # absmax read + convert read + write
        rc = lib.mdl_split_image(_ptr(x2d), x2d.stride(0), rows, K, _ptr(data), K * 4, pad_rows, _ptr(scale), _stream())
    _native.check(rc, "mdl_split_image")
    return SplitImage(data, scale, rows, K)


def split_image_rows(x2d, pad_rows=0):
    """Row-scaled image of x2d (mdl_split_image_rows): -> (SplitImage with common scale 1, row_inv [rows] = 1 / the row scales).  For the
    tensor whose rows the caller controls (the patch features): one outlier patch does not cost the other patches their low bits."""
    _require(x2d, "x")
    lib = _native.lib()
    rows, K = x2d.shape
    data = torch.empty(rows + pad_rows, K, device=x2d.device, dtype=torch.float32)
    row_inv = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    # scale = NULL: nothing consumes the tensor-wide maximum of a row-scaled image, and collecting it is a memset, a scale launch and one
    # atomic per wave on a single address (75 us of a 95-us launch at 30,000 rows: every wave arrives at once)
    with _timed("split_image", ("byte", 8.0 * rows * K)):   # one HBM read + write
        rc = lib.mdl_split_image_rows(_ptr(x2d), x2d.stride(0), rows, K, _ptr(data), K * 4, pad_rows, _ptr(row_inv), None, _stream())
    _native.check(rc, "mdl_split_image_rows")
    return SplitImage(data, _unit_scale(x2d.device), rows, K), row_inv


_UNIT_SCALE = {}


def _unit_scale(device):
    """The constant {1, 0} on `device`: the common scale of a row-scaled image (one tensor per device, never written)."""
    key = str(device)
    if key not in _UNIT_SCALE:
        _UNIT_SCALE[key] = torch.tensor([1.0, 0.0], device=device, dtype=torch.float32)
    return _UNIT_SCALE[key]


def weight_image(W) -> SplitImage:
    """Row-scaled image of a weight matrix W [N, K] (one power of two per output channel) in ONE launch -- instead of the memset, absmax,
    scale and convert launches of split_image, every forward and backward of every Linear.  As the B operand of split_gemm_nt its row
    factors become per-output-column factors of the product (b_col_mul)."""
    _require(W, "weight")
    lib = _native.lib()
    rows, K = W.shape
    data = torch.empty(rows, K, device=W.device, dtype=torch.float32)
    row_inv = torch.empty(rows, device=W.device, dtype=torch.float32)
    rc = lib.mdl_split_image_rows(_ptr(W), W.stride(0), rows, K, _ptr(data), K * 4, 0, _ptr(row_inv), None, _stream())
    _native.check(rc, "mdl_split_image_rows")
    return SplitImage(data, _unit_scale(W.device), rows, K, row_inv)


def split_tile_absmax(x2d, chunks=False):
    """max |x| of every block of 256 rows (the row gate of split_gemm_nt); chunks=True: also of every 32 rows (split_gemm_tn's
    b_chunk_max) -> (gate, chunk_max)."""
    lib = _native.lib()
    rows, K = x2d.shape
    gate = torch.empty((rows + 255) // 256, device=x2d.device, dtype=torch.float32)
    cm = torch.empty((rows + 31) // 32, device=x2d.device, dtype=torch.float32) if chunks else None
    _native.check(lib.mdl_split_tile_absmax(_ptr(x2d), x2d.stride(0), rows, K, _ptr(gate), _ptr(cm), _stream()), "mdl_split_tile_absmax")
    return (gate, cm) if chunks else gate


def split_gemm_nt(A: SplitImage, B: SplitImage, bias=None, out=None, accumulate=False, absmax_out=None, name="split_nt", row_gate=None,
                  a_row_mul=None, terms=3):
    """C [A.rows, B.rows] (+)= A B^T (+ bias) on two images with the same K.  row_gate (accumulate mode only): per-256-row maxima of
    the tensor A is the image of; output tiles of all-zero A rows are skipped.  a_row_mul [A.rows]: per-row factor applied to the
    product (row_inv of a row-scaled A image).  terms=2: B enters with its hi plane only (GRAD_TERMS)."""
    lib = _native.lib()
    M, N, K = A.rows, B.rows, A.K
    if B.K != K:
        raise ValueError("split_gemm_nt: contraction lengths differ")
    if a_row_mul is None:
        a_row_mul = A.row_inv
    C = out if out is not None else torch.empty(M, N, device=A.data.device, dtype=torch.float32)
    with _timed(name, ("flop", 2.0 * M * N * K)):
        rc = lib.mdl_split_gemm_nt(_ptr(A.data), K * 4, _ptr(A.scale), _ptr(B.data), K * 4, _ptr(B.scale), _ptr(C), C.stride(0), M, N, K,
                                   _ptr(bias), int(accumulate), _ptr(absmax_out), _ptr(row_gate), _ptr(a_row_mul), _ptr(B.row_inv), int(terms),
                                   _stream())
    _native.check(rc, "mdl_split_gemm_nt")
    return C


def split_gemm_nt_group_bias(A: SplitImage, B: SplitImage, group_bias, row_group, bias=None, a_row_mul=None, name="split_nt"):
    """split_gemm_nt with a bias row per GROUP of output rows: C[m] = A[m] B^T (+ bias) + group_bias[row_group[m]] (mdl_split_gemm_nt_group_bias;
    group_bias [G, B.rows] fp32, row_group int32 [A.rows])."""
    lib = _native.lib()
    M, N, K = A.rows, B.rows, A.K
    if B.K != K:
        raise ValueError("split_gemm_nt_group_bias: contraction lengths differ")
    _require(group_bias, "group_bias")
    _require(row_group, "row_group", torch.int32)
    if group_bias.dim() != 2 or group_bias.shape[1] != N or row_group.numel() != M:
        raise ValueError("split_gemm_nt_group_bias: group_bias must be [G, %d] and row_group [%d]" % (N, M))
    if a_row_mul is None:
        a_row_mul = A.row_inv
    C = torch.empty(M, N, device=A.data.device, dtype=torch.float32)
    with _timed(name, ("flop", 2.0 * M * N * K)):
        rc = lib.mdl_split_gemm_nt_group_bias(_ptr(A.data), K * 4, _ptr(A.scale), _ptr(B.data), K * 4, _ptr(B.scale), _ptr(C), C.stride(0), M, N, K,
                                              _ptr(bias), _ptr(a_row_mul), _ptr(B.row_inv), _ptr(group_bias), _ptr(row_group), 3, _stream())
    _native.check(rc, "mdl_split_gemm_nt_group_bias")
    return C


def split_gemm_tn(A: SplitImage, B: SplitImage, name="split_tn", b_chunk_max=None, terms=3):
    """out [B.K, A.K] = B^T A summed over the rows (tokens) of the two images; B must carry >= 32 zero pad rows.  b_chunk_max: the
    per-32-row maxima (split_tile_absmax(x, chunks=True)) of the tensor B is the image of -- its all-zero chunks are skipped."""
    lib = _native.lib()
    T, Mi, N = A.rows, A.K, B.K
    if B.row_inv is not None:
        raise ValueError("split_gemm_tn: a row-scaled image cannot be the B operand of a contraction over its rows")
    if B.rows != T or B.data.shape[0] < T + 32:
        raise ValueError("split_gemm_tn: images need the same number of rows and B 32 zero pad rows")
    out = torch.empty(N, Mi, device=A.data.device, dtype=torch.float32)
    ws = _ws(lib.mdl_split_gemm_tn_ws_bytes(T, Mi, N), A.data.device)
    with _timed(name, ("flop", 2.0 * T * Mi * N)):
        rc = lib.mdl_split_gemm_tn(_ptr(A.data), Mi * 4, _ptr(A.scale), Mi, _ptr(B.data), N * 4, _ptr(B.scale), N, _ptr(out), T, _ptr(b_chunk_max),
                                   _ptr(ws), int(terms), _stream())
    _native.check(rc, "mdl_split_gemm_tn")
    return out


def _split_gate(E2d) -> bool:
    """The gate (A2) contractions take the split engine for fp32 token embeddings in the 'split' GEMM mode."""
    return GEMM_MODE == "split" and E2d.dtype == torch.float32 and E2d.shape[0] > 0


def split_linear_supported(T, N, K) -> bool:
    """Geometries the split engine takes for a Linear [T,K] x [N,K]: fwd K % 32, N % 4; dX: N % 32 (contraction), K % 4; dW: both % 32."""
    return T > 256 and K % 32 == 0 and N % 32 == 0


class SplitLinearFn(torch.autograd.Function):
    """LinearFn in the split GEMM mode: Y = X W^T (+ bias) with fp32 inputs / outputs, the contraction on the split-fp16 engine.
    x may be given as a ready SplitImage (`x_img`, from a producer kernel that writes images directly); otherwise it is built here."""

    @staticmethod
    def forward(ctx, x, W, bias):
        _require(x, "x")
        _require(W, "weight")
        if bias is not None:
            _require(bias, "bias")
        xi = split_image(x)
        y = split_gemm_nt(xi, weight_image(W), bias, name="linear_fwd")
        ctx.save_for_backward(W, xi.data, xi.scale)     # (the image is an ordinary saved tensor: hooks, retain_graph, version checks)
        ctx.geom = (xi.rows, xi.K)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        W, xdata, xscale = ctx.saved_tensors
        xi = SplitImage(xdata, xscale, *ctx.geom)
        dy = dy.float().contiguous()
        dyi = split_image(dy, pad_rows=32)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = split_gemm_nt(dyi, weight_image(W.t().contiguous()), name="linear_bwd", terms=GRAD_TERMS)
        dW = split_gemm_tn(xi, dyi, name="linear_bwd", terms=GRAD_TERMS)
        db = dy.sum(0) if ctx.has_bias else None
        return dx, dW, db


# max |t| of a gradient tensor, published by the kernel that produced it (epilogue atomics) for the consumer that needs a bound on it.
# An entry binds to the CONTENTS of the buffer, not just its address: it keeps the producer's tensor alive (the address cannot be
# recycled for another tensor, and the autograd engine cannot steal the buffer to sum a second gradient into it in place -- it
# allocates the sum elsewhere, which misses here) and records the version counter the tensor had when the kernel wrote it (views share
# the counter: an in-place update by a hook or another consumer shows as a newer version).  A miss or a stale entry returns None and the
# consumer takes its own one-pass absmax (the dy_absmax = NULL path of mdl_ln_gelu_drop_bwd_split).
# The strong reference lives only as long as the backward pass that created it: the first entry of a pass queues an engine callback
# that empties the table when the outermost backward returns, so a consumer that never runs (frozen pre_attn, a hook that replaces the
# gradient, an exception) cannot pin a multi-GB gradient buffer until the next encoder forward.
_ABSMAX = {}
_ABSMAX_LOCK = threading.Lock()
_ABSMAX_CB = [False]


def _absmax_end_of_backward():
    with _ABSMAX_LOCK:
        _ABSMAX.clear()
        _ABSMAX_CB[0] = False


def _put_absmax(t, amax):
    with _ABSMAX_LOCK:
        if len(_ABSMAX) > 8:
            _ABSMAX.clear()
        _ABSMAX[(t.data_ptr(), t.numel())] = (t, t._version, amax)
        if not _ABSMAX_CB[0]:
            try:      # (only callable while the engine is running a backward -- which is where every producer lives)
                torch.autograd.Variable._execution_engine.queue_callback(_absmax_end_of_backward)
                _ABSMAX_CB[0] = True
            except RuntimeError:
                pass


def _take_absmax(t):
    with _ABSMAX_LOCK:
        e = _ABSMAX.pop((t.data_ptr(), t.numel()), None)
    if e is None:
        return None
    src, version, amax = e
    fresh = (src._version == version and t._version == version
             and src.untyped_storage().data_ptr() == t.untyped_storage().data_ptr())
    return amax if fresh else None


def _clear_absmax():
    # Start of an encoder forward.  Also re-arms the end-of-backward callback: when a backward raises, the engine skips its final
    # callbacks, so the flag set by _put_absmax would otherwise stay True for the life of the process.
    with _ABSMAX_LOCK:
        _ABSMAX.clear()
        _ABSMAX_CB[0] = False


class PreAttnBlockFn(torch.autograd.Function):
    """One block of the pre-attention MLP -- Linear -> LayerNorm -> GELU -> Dropout (Model.py:351-354, :355-358, :359-362) -- as ONE
    autograd node on the split engine.  The activations between the blocks exist as split images only (consumed by contractions):
    `x` is either the fp32 input of the first block (x_scale None) or the previous block's image (x_scale = its scale); the block
    returns (image [T,N] as an opaque float32 tensor, scale [2], fp32 output or an empty tensor).  Gradients travel as ordinary fp32
    tensors of the images' shape; in backward the LayerNorm kernel writes d(pre-LN) as an image straight away and the dX / dW
    contractions read it -- no conversion pass anywhere (scales come from rigorous bounds, include/madeleine_amd.h)."""

    @staticmethod
    def forward(ctx, x, x_scale, W, lin_bias, gamma, beta, eps, p_drop, seed, keep, want_fp32, gbias=None, row_group=None, cu_groups=None):
        # gbias [G, N] + row_group int32 [T] + cu_groups int64 [G + 1] (first block only, round 5): a bias row per GROUP of rows (= bag), rows of
        # a group contiguous -- the stain-encoding columns of MADELEINE's first Linear folded out of the contraction:
        # [x | e_g] W^T = x Wx^T + e_g We^T (Model.py:125-132, :351).  The concat [T, D + 32] never exists; the backward returns d(gbias).
        _require(x, "x")
        for t_, n_ in ((W, "weight"), (gamma, "gamma"), (beta, "beta")):
            _require(t_, n_)
        if lin_bias is not None:
            _require(lin_bias, "bias")
        lib = _native.lib()
        T, K = x.shape
        N = W.shape[0]
        dev = x.device
        if x_scale is None:
            _clear_absmax()      # first block of an encoder forward: no published maximum outlives a step
        # first block: x = the caller's patch features -> ROW-scaled image (an outlier patch must not cost the others their low bits;
        # nn.Linear in fp32 has no coupling between rows); later blocks: the previous block's LayerNorm output, rows of one magnitude
        row_inv = None
        if x_scale is not None:
            xi = SplitImage(x, x_scale, T, K)
        else:
            xi, row_inv = split_image_rows(x)
        if gbias is not None:
            _require(cu_groups, "cu_groups", torch.int64)
            y = split_gemm_nt_group_bias(xi, weight_image(W), gbias.contiguous(), row_group, a_row_mul=row_inv, name="linear_fwd")
        else:
            y = split_gemm_nt(xi, weight_image(W), name="linear_fwd", a_row_mul=row_inv)   # pre-LN values (the Linear's bias is added by the LN kernel)
        img = torch.empty(T, N, device=dev, dtype=torch.float32)
        scale = torch.empty(2, device=dev, dtype=torch.float32)
        out = torch.empty(T, N, device=dev, dtype=torch.float32) if want_fp32 else None
        mean = torch.empty(T, device=dev, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        # max_r rstd[r] row_inv[r], a factor of the backward's image bound -- only when a backward can follow (inference: not even the
        # one atomic per workgroup)
        needs_bwd = any(ctx.needs_input_grad)
        rstd_max = torch.empty(1, device=dev, dtype=torch.float32) if needs_bwd else None
        with _timed("ln_gelu_drop_fwd", ("byte", (3.0 if want_fp32 else 2.0) * T * N * 4)):
            rc = lib.mdl_ln_gelu_drop_fwd_split(_ptr(y), _ptr(lin_bias), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(img), _ptr(scale),
                                                _ptr(mean), _ptr(rstd), T, N, float(eps), float(p_drop), int(seed), _ptr(keep),
                                                _ptr(row_inv), _ptr(rstd_max), _stream())
        if rc == -3:
            raise NotImplementedError("fused LayerNorm-GELU-Dropout supports widths 256/512/1024/2048/4096 (got %d)" % N)
        _native.check(rc, "mdl_ln_gelu_drop_fwd_split")
        ctx.save_for_backward(xi.data, xi.scale, W, y, gamma, beta, mean, rstd, lin_bias if lin_bias is not None else torch.empty(0),
                              row_inv if row_inv is not None else torch.empty(0), rstd_max if rstd_max is not None else torch.empty(0))
        ctx.cfg = (float(p_drop), int(seed), keep, lin_bias is not None, bool(want_fp32), T, K, N, x_scale is not None)
        ctx.groups = None if gbias is None else (cu_groups, int(gbias.shape[0]))
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(scale)
        if want_fp32:
            ctx.mark_non_differentiable(img)      # the gradient arrives on the fp32 output
            return img, scale, out
        empty = x.new_empty(0)
        ctx.mark_non_differentiable(empty)
        return img, scale, empty

    @staticmethod
    def backward(ctx, d_img, _d_scale, d_out):
        xdata, xscale, W, y, gamma, beta, mean, rstd, lin_bias, row_inv, rstd_max = ctx.saved_tensors
        p_drop, seed, keep, has_bias, want_fp32, T, K, N, x_is_image = ctx.cfg
        lin_bias = lin_bias if has_bias else None
        rstd_max = rstd_max if rstd_max.numel() else None
        # first block (row-scaled input image): the gradient image carries row_inv[r] dx[r][:], so that the row factors of the two images
        # cancel inside the dW contraction over rows
        row_inv = None if x_is_image else row_inv
        dy = d_out if want_fp32 else d_img
        if dy is None:
            dy = torch.zeros(T, N, device=y.device, dtype=torch.float32)
        dy = dy.float().contiguous().view(T, N)
        lib = _native.lib()
        dev = y.device
        dximg = torch.empty(T + 32, N, device=dev, dtype=torch.float32)
        dxscale = torch.empty(2, device=dev, dtype=torch.float32)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        dbias = torch.empty_like(lin_bias) if has_bias else None
        amax = _take_absmax(dy)
        d_gbias = None
        if ctx.groups is not None:
            cu_groups, G = ctx.groups
            d_gbias = torch.empty(G, N, device=dev, dtype=torch.float32)
            ws = _ws(lib.mdl_ln_gelu_drop_bwd_groups_ws_bytes(T, N, G), dev)
            with _timed("ln_gelu_drop_bwd", ("byte", (3.0 if amax is not None else 4.0) * T * N * 4)):
                rc = lib.mdl_ln_gelu_drop_bwd_split_groups(_ptr(y), _ptr(lin_bias), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dy),
                                                           _ptr(amax), _ptr(dximg), _ptr(dxscale), _ptr(dg), _ptr(db), _ptr(dbias), T, N, p_drop, seed,
                                                           _ptr(keep), _ptr(row_inv), _ptr(rstd_max), _ptr(cu_groups), G, _ptr(d_gbias), _ptr(ws),
                                                           _stream())
            _native.check(rc, "mdl_ln_gelu_drop_bwd_split_groups")
        else:
            ws = _ws(lib.mdl_ln_gelu_drop_bwd_ws_bytes(T, N), dev)
            with _timed("ln_gelu_drop_bwd", ("byte", (3.0 if amax is not None else 4.0) * T * N * 4)):
                rc = lib.mdl_ln_gelu_drop_bwd_split(_ptr(y), _ptr(lin_bias), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dy), _ptr(amax),
                                                    _ptr(dximg), _ptr(dxscale), _ptr(dg), _ptr(db), _ptr(dbias), T, N, p_drop, seed, _ptr(keep),
                                                    _ptr(row_inv), _ptr(rstd_max), _ptr(ws), _stream())
            _native.check(rc, "mdl_ln_gelu_drop_bwd_split")
        dyi = SplitImage(dximg, dxscale, T, N)
        dx = None
        if ctx.needs_input_grad[0]:
            am = torch.zeros(1, device=dev, dtype=torch.float32)
            dxi = dyi
            if row_inv is not None:
                # d(patch features) is wanted (stain-encoding tokens concatenated to the bags, Model.py:132; a caller differentiating
                # w.r.t. its features): the image above carries the dW pairing's row factors, under which a row's own gradient can sit
                # far below the image's scale.  A second pass writes d(pre-LN) under the common scale for this product alone.
                dximg2 = torch.empty(T + 32, N, device=dev, dtype=torch.float32)
                dxscale2 = torch.empty(2, device=dev, dtype=torch.float32)
                dg2, db2 = torch.empty_like(gamma), torch.empty_like(beta)
                with _timed("ln_gelu_drop_bwd", ("byte", 3.0 * T * N * 4)):
                    rc = lib.mdl_ln_gelu_drop_bwd_split(_ptr(y), _ptr(lin_bias), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dy), _ptr(amax),
                                                        _ptr(dximg2), _ptr(dxscale2), _ptr(dg2), _ptr(db2), None, T, N, p_drop, seed, _ptr(keep),
                                                        None, None, _ptr(ws), _stream())   # (no row factors: its own pass over rstd)
                _native.check(rc, "mdl_ln_gelu_drop_bwd_split")
                dxi = SplitImage(dximg2, dxscale2, T, N)
            dx = split_gemm_nt(dxi, weight_image(W.t().contiguous()), absmax_out=am, name="linear_bwd", terms=GRAD_TERMS)
            if x_is_image:       # the consumer is the previous block's LayerNorm backward (this node's input was its image)
                _put_absmax(dx, am)
        dW = split_gemm_tn(SplitImage(xdata, xscale, T, K), dyi, name="linear_bwd", terms=GRAD_TERMS)
        return dx, None, dW, dbias, dg, db, None, None, None, None, None, d_gbias, None, None


def preattn_block(x, x_scale, W, lin_bias, gamma, beta, eps=1e-5, p_drop=0.0, seed=0, keep=None, want_fp32=False, group_bias=None,
                  row_group=None, cu_groups=None):
    return PreAttnBlockFn.apply(x, x_scale, W.contiguous(), None if lin_bias is None else lin_bias.contiguous(), gamma.contiguous(),
                                beta.contiguous(), float(eps), float(p_drop), int(seed), keep, bool(want_fp32), group_bias, row_group, cu_groups)


def preattn_split_supported(x2d, K) -> bool:
    """x2d: the [T, K] bags as the caller holds them (any floating dtype: they are widened to fp32 for the first block, no copy is made
    to answer this)."""
    return GEMM_MODE == "split" and x2d.is_cuda and x2d.is_floating_point() and x2d.shape[0] > 256 and K % 32 == 0


def linear_fwd_raw(x2d, W, bias):
    """Y = X W^T (+ bias) through mdl_linear_fwd / mdl_linear_fwd_bf16 (by the storage type of x2d)."""
    lib = _native.lib()
    T, K = x2d.shape
    N = W.shape[0]
    sfx = _sfx(x2d)
    y = torch.empty(T, N, device=x2d.device, dtype=x2d.dtype)
    ws = _ws(getattr(lib, "mdl_linear_fwd%s_ws_bytes" % sfx)(T, N, K), x2d.device)
    with _timed("linear_fwd", ("flop", 2.0 * T * N * K)):
        rc = getattr(lib, "mdl_linear_fwd" + sfx)(_ptr(x2d), x2d.stride(0), _ptr(W), _ptr(bias), _ptr(y), N, T, N, K, _ptr(ws), _stream())
    _native.check(rc, "mdl_linear_fwd" + sfx)
    return y


def linear_bwd_raw(x2d, W, dy, dx, want_dbias):
    """(dW, dbias) of the Linear; dX is written into `dx` ([T,K], may be None)."""
    lib = _native.lib()
    T, K = x2d.shape
    N = W.shape[0]
    sfx = _sfx(x2d)
    dW = torch.empty_like(W)
    db = torch.empty(N, device=x2d.device, dtype=torch.float32) if want_dbias else None
    ws = _ws(getattr(lib, "mdl_linear_bwd%s_ws_bytes" % sfx)(T, N, K), x2d.device)
    with _timed("linear_bwd", ("flop", 2.0 * T * N * K * (2 if dx is not None else 1))):
        rc = getattr(lib, "mdl_linear_bwd" + sfx)(_ptr(x2d), x2d.stride(0), _ptr(W), _ptr(dy), N, _ptr(dx), K, _ptr(dW), _ptr(db), T, N, K,
                                                  _ptr(ws), _stream())
    _native.check(rc, "mdl_linear_bwd" + sfx)
    return dW, db


def _bag_geometry(E, cu_seqlens, max_len):
    """E is [n_bags,N,C] (dense) or [T,C] with cu_seqlens int64 [n_bags+1] (ragged)."""
    if cu_seqlens is None:
        if E.dim() != 3:
            raise ValueError("dense bags must be [n_bags, N, H*512]")
        n_bags, N = E.shape[0], E.shape[1]
        return n_bags, N, N, E.reshape(n_bags * N, E.shape[2])
    if E.dim() != 2:
        raise ValueError("ragged bags must be packed [T, H*512] with cu_seqlens")
    _require(cu_seqlens, "cu_seqlens", torch.int64)
    if max_len is None:
        raise ValueError("max_len is required with cu_seqlens (avoids a device sync)")
    return cu_seqlens.numel() - 1, 0, int(max_len), E


# --------------------------------------------------------------------------------------------------
# A2: gated attention scores
# --------------------------------------------------------------------------------------------------
class GateScoresFn(torch.autograd.Function):
    """scores[T,H] = gated-attention raw scores of all heads (abmil.py:41-52 per head)."""

    @staticmethod
    def forward(ctx, E2d, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b):
        _require_act(E2d, "E")
        for t, n in ((Wa, "Wa"), (ba, "ba"), (Wb, "Wb"), (bb, "bb"), (wc, "wc"), (bc, "bc")):
            _require(t, n)
        need = any(ctx.needs_input_grad[:7])
        Ei = None
        if _split_gate(E2d):
            Ei = split_image(E2d)
            scores, act_a, act_b = gate_fwd_split_raw(Ei, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, need)
        else:
            scores, act_a, act_b = gate_fwd_raw(E2d, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, need)
        if need:
            none = torch.empty(0)
            ctx.save_for_backward(E2d, Wa, Wb, wc, act_a, act_b, Ei.data if Ei is not None else none, Ei.scale if Ei is not None else none)
            ctx.drop = (p_drop, seed, keep_a, keep_b)
            ctx.has_image = Ei is not None
        return scores

    @staticmethod
    def backward(ctx, d_scores):
        E2d, Wa, Wb, wc, act_a, act_b, Eidata, Eiscale = ctx.saved_tensors
        p_drop, seed, keep_a, keep_b = ctx.drop
        d_scores = d_scores.float().contiguous()
        dE = torch.empty_like(E2d)
        if ctx.has_image:
            Ei = SplitImage(Eidata, Eiscale, E2d.shape[0], E2d.shape[1])
            dWa, dWb, dba, dbb, dwc, dbc = attnpool_bwd_split_raw(Ei, Wa, Wb, wc, act_a, act_b, d_scores, dE, p_drop, seed, keep_a,
                                                                  keep_b, None, None, None, None, None, 0)
        else:
            dWa, dWb, dba, dbb, dwc, dbc = gate_bwd_raw(E2d, Wa, Wb, wc, act_a, act_b, d_scores, dE, 0, p_drop, seed, keep_a,
                                                        keep_b)
        return dE, dWa, dba, dWb, dbb, dwc, dbc, None, None, None, None


# --------------------------------------------------------------------------------------------------
# A3: softmax over patches + pooling
# --------------------------------------------------------------------------------------------------
class SoftmaxPoolFn(torch.autograd.Function):
    """pooled[b,c,:] = sum_t softmax_t(scores[b,:,c]) E[b,t,c,:]   (abmil.py:55 + Model.py:416-417)."""

    @staticmethod
    def forward(ctx, E, scores, cu_seqlens, max_len):
        _require_act(E, "E")
        _require(scores, "scores")
        n_bags, N, max_len, E2d = _bag_geometry(E, cu_seqlens, max_len)
        s2d = scores.reshape(E2d.shape[0], -1)
        pooled, m, l = pool_fwd_raw(E2d, s2d, n_bags, N, cu_seqlens, max_len)
        ctx.save_for_backward(E2d, s2d, pooled, m, l, cu_seqlens if cu_seqlens is not None else torch.empty(0))
        ctx.geom = (n_bags, N, max_len, cu_seqlens is not None, E.shape, scores.shape)
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        E2d, s2d, pooled, m, l, cu = ctx.saved_tensors
        n_bags, N, max_len, ragged, e_shape, s_shape = ctx.geom
        cu = cu if ragged else None
        dE = torch.empty_like(E2d)
        ds = torch.empty_like(s2d)
        pool_bwd_raw(E2d, s2d, pooled, m, l, d_pooled.float().contiguous(), dE, 0, ds, 0, n_bags, N, cu, max_len)
        return dE.view(e_shape), ds.view(s_shape), None, None


class WeightedPoolFn(torch.autograd.Function):
    """pooled[b,c,:] = sum_t weights[b,t,c] E[b,t,c,:] with the weights taken as they are (no softmax): the relu / leaky_relu /
    sigmoid attention activations of abmil.py:56-61 pooled as Model.py:416-417 does."""

    @staticmethod
    def forward(ctx, E, weights, cu_seqlens, max_len):
        _require_act(E, "E")
        _require(weights, "weights")
        n_bags, N, max_len, E2d = _bag_geometry(E, cu_seqlens, max_len)
        w2d = weights.reshape(E2d.shape[0], -1)
        lib = _native.lib()
        H = w2d.shape[-1]
        pooled = torch.empty(n_bags, H * HID, device=E2d.device, dtype=torch.float32)
        scratch = torch.empty(2, n_bags, H, device=E2d.device, dtype=torch.float32)
        ws = _ws(lib.mdl_abmil_pool_ws_bytes(n_bags, max_len, H), E2d.device)
        with _timed("pool_fwd"):
            rc = getattr(lib, "mdl_abmil_wpool_fwd" + _sfx(E2d))(_ptr(E2d), E2d.stride(0), _ptr(w2d), _ptr(pooled), _ptr(scratch[0]),
                                                                 _ptr(scratch[1]), n_bags, N, _ptr(cu_seqlens), max_len, H, _ptr(ws),
                                                                 _stream())
        _native.check(rc, "mdl_abmil_wpool_fwd")
        ctx.save_for_backward(E2d, w2d, cu_seqlens if cu_seqlens is not None else torch.empty(0))
        ctx.geom = (n_bags, N, max_len, cu_seqlens is not None, E.shape, weights.shape)
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        E2d, w2d, cu = ctx.saved_tensors
        n_bags, N, max_len, ragged, e_shape, w_shape = ctx.geom
        cu = cu if ragged else None
        dE = torch.empty_like(E2d)
        dw = torch.empty_like(w2d)
        lib = _native.lib()
        with _timed("pool_bwd"):
            rc = getattr(lib, "mdl_abmil_wpool_bwd" + _sfx(E2d))(_ptr(E2d), E2d.stride(0), _ptr(w2d), _ptr(d_pooled.float().contiguous()),
                                                                 _ptr(dE), 0, _ptr(dw), n_bags, N, _ptr(cu), max_len, w2d.shape[-1],
                                                                 _stream())
        _native.check(rc, "mdl_abmil_wpool_bwd")
        return dE.view(e_shape), dw.view(w_shape), None, None


def weighted_pool(E, weights, cu_seqlens=None, max_len=None):
    return WeightedPoolFn.apply(E, weights.contiguous(), cu_seqlens, max_len)


# --------------------------------------------------------------------------------------------------
# A2 + A3 chained inside one autograd node: the pooling backward only produces d_scores, and its dE term
# w * d_pooled is added in the gate backward's dX epilogue -- dE is written exactly once.
# --------------------------------------------------------------------------------------------------
class AttnPoolFn(torch.autograd.Function):
    """(pooled [n_bags,(1+V,)H*512], raw scores [T,H], token projections [T,P]) = multi-head gated-ABMIL pooling (Model.py:406-417)
    and, with `views` = V int32 token-index lists (dense bags only), the V re-softmaxed sub-bag poolings of Model.py:419-440.
    With Wtok [P, H*512] (+ btok) the token_projector Linear (Model.py:140) -- the other consumer of E -- is part of the node: its dX
    is written into dE first and the gate dX epilogue accumulates onto it, so the two gradients of E are never summed by a separate
    3 x |E| elementwise pass (3.8 ms per config-3 step); without it the third output is an empty tensor."""

    @staticmethod
    def forward(ctx, E, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, cu_seqlens, max_len, Wtok, btok, Eimg, Escale, e_only_image,
                *views):
        # e_only_image: E IS the image tensor (Eimg is E, an opaque float32 [.., H*512] tensor written by the last pre_attn block's
        # LayerNorm kernel) -- no fp32 copy of E exists; the pooling kernels read the image, dE is the gradient of that tensor
        _require_act(E, "E")
        for t, n in ((Wa, "Wa"), (ba, "ba"), (Wb, "Wb"), (bb, "bb"), (wc, "wc"), (bc, "bc")):
            _require(t, n)
        n_bags, N, max_len, E2d = _bag_geometry(E, cu_seqlens, max_len)
        if views and cu_seqlens is not None:
            raise NotImplementedError("token-index views are defined on dense bags (the reference's n_views path stacks equal-N bags)")
        for v in views:
            _require(v, "view token indices", torch.int32)
        ctx.set_materialize_grads(False)
        need = any(ctx.needs_input_grad[:7]) or any(ctx.needs_input_grad[13:15])   # (inputs 15, 16 = the image of E: no gradient)
        Ei = None
        if e_only_image and (Eimg is None or views or not _split_gate(E2d)
                             or (Wtok is not None and not split_linear_supported(E2d.shape[0], Wtok.shape[0], Wtok.shape[1]))):
            raise RuntimeError("attn_pool: an image-only E needs the split GEMM mode, no token views and a token projection the split "
                               "engine serves")
        recompute = need and GATE_RECOMPUTE
        if _split_gate(E2d):
            # the image of E: written by the producing LayerNorm kernel (Eimg / Escale), else built here (3 passes over E)
            Ei = SplitImage(Eimg, Escale, E2d.shape[0], E2d.shape[1]) if Eimg is not None else split_image(E2d)
            scores, act_a, act_b = gate_fwd_split_raw(Ei, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, need and not recompute)
        else:
            scores, act_a, act_b = gate_fwd_raw(E2d, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, need and not recompute)
        pooled, m, l = (pool_fwd_img_raw(Ei, scores, n_bags, N, cu_seqlens, max_len) if e_only_image
                        else pool_fwd_raw(E2d, scores, n_bags, N, cu_seqlens, max_len))
        vstate = [pool_view_fwd_raw(E2d, scores, n_bags, N, v) for v in views]
        if Wtok is not None:
            _require(Wtok, "token_projector weight")
            if btok is not None:
                _require(btok, "token_projector bias")
            if Ei is not None and split_linear_supported(E2d.shape[0], Wtok.shape[0], Wtok.shape[1]):
                # split engine on the image of E the gate forward used (N = 128: half of the 256-wide tile idles, still faster than
                # the fp32 tall tile)
                tok = split_gemm_nt(Ei, weight_image(Wtok), btok, name="linear_fwd")
            else:
                if not linear_supported(E2d, Wtok):   # a clear message instead of a kernel return code
                    raise NotImplementedError("attn_pool: token projection %s on %s %s rows is outside the HIP Linear kernels' geometries "
                                              "(functional.linear_supported)" % (tuple(Wtok.shape), E2d.shape[0], E2d.dtype))
                tok = linear_fwd_raw(E2d, Wtok, btok)
        else:
            tok = E2d.new_empty(0)
            ctx.mark_non_differentiable(tok)
        if need:
            flat = [t for st in vstate for t in st]
            none = torch.empty(0)
            # (the image of E travels as saved tensors like everything else: saved-tensor hooks, retain_graph and version checks apply)
            ctx.save_for_backward(E2d, Wa, Wb, wc, act_a if not recompute else none, act_b if not recompute else none, scores, pooled, m, l,
                                  cu_seqlens if cu_seqlens is not None else none, Wtok if Wtok is not None else none,
                                  Ei.data if Ei is not None else none, Ei.scale if Ei is not None else none,
                                  ba if recompute else none, bb if recompute else none, bc if recompute else none, *views, *flat)
            ctx.recompute = recompute
            ctx.cfg = (p_drop, seed, keep_a, keep_b, n_bags, N, max_len, cu_seqlens is not None, E.shape, len(views),
                       Wtok is not None, btok is not None)
            ctx.e_only_image = bool(e_only_image)
            ctx.has_image = Ei is not None
        if views:
            pooled = torch.stack([pooled] + [st[0] for st in vstate], dim=1)
        return pooled, scores, tok

    @staticmethod
    def backward(ctx, d_pooled, d_scores_in, d_tok):
        p_drop, seed, keep_a, keep_b, n_bags, N, max_len, ragged, e_shape, V, has_tok, has_btok = ctx.cfg
        saved = ctx.saved_tensors
        E2d, Wa, Wb, wc, act_a, act_b, scores, pooled, m, l, cu, Wtok, Eidata, Eiscale, ba, bb, bc = saved[:17]
        views, vflat = saved[17:17 + V], saved[17 + V:]
        Ei = SplitImage(Eidata, Eiscale, E2d.shape[0], E2d.shape[1]) if ctx.has_image else None
        if ctx.recompute:   # rebuild the activations: the same kernel, seed and masks as the forward -> the same bits
            if Ei is not None:
                _s, act_a, act_b = gate_fwd_split_raw(Ei, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, True)
            else:
                _s, act_a, act_b = gate_fwd_raw(E2d, Wa, ba, Wb, bb, wc, bc, p_drop, seed, keep_a, keep_b, True)
        cu = cu if ragged else None
        dE = torch.empty_like(E2d)
        if d_scores_in is not None:
            ds = d_scores_in.float().contiguous().clone()
            acc_s = 1
        else:
            ds = torch.empty_like(scores)
            acc_s = 0
        if d_pooled is None:
            d_pooled = torch.zeros(n_bags, 1 + V, pooled.shape[-1], device=pooled.device) if V else torch.zeros_like(pooled)
        d_pooled = d_pooled.float().contiguous()
        d_main = d_pooled[:, 0].contiguous() if V else d_pooled
        # the other consumer of E, the token_projector: its dX is accumulated into the same dE buffer (no separate gradient tensor, no
        # add pass).  Split engine: AFTER the gate backward, and only on the 256-token tiles where d_tok is not identically zero (the
        # local loss reads the first <= 256 tokens of a bag, so d_tok is zero on ~94 % of the tiles at N = 4096); the other engines:
        # before it (the gate dX epilogue then accumulates).
        dWtok = dbtok = None
        acc_e = 0
        tok_after = has_tok and d_tok is not None and Ei is not None and split_linear_supported(E2d.shape[0], Wtok.shape[0], Wtok.shape[1])
        if has_tok and d_tok is not None and not tok_after:
            dWtok, dbtok = linear_bwd_raw(E2d, Wtok, d_tok.to(E2d.dtype).contiguous(), dE, has_btok)
            acc_e = 1
        # scores-only pooling backward (one read of E), then the gate backward whose dX epilogue adds the pooling term
        if ctx.e_only_image:
            pool_dscores_img_raw(Ei, scores, pooled, m, l, d_main, ds, acc_s, n_bags, N, cu, max_len)
        else:
            pool_bwd_raw(E2d, scores, pooled, m, l, d_main, None, 0, ds, acc_s, n_bags, N, cu, max_len)
        for i in range(V):   # the views' score gradients must be in ds before the gate backward consumes it
            vp, vm, vl = vflat[3 * i:3 * i + 3]
            pool_view_bwd_raw(E2d, scores, vp, vm, vl, d_pooled[:, 1 + i].contiguous(), None, ds, n_bags, N, views[i])
        row_bag = None
        if ragged:   # bag index of every packed token row, on the device (no sync)
            row_bag = torch.searchsorted(cu[1:].contiguous(), torch.arange(E2d.shape[0], device=E2d.device), right=True).to(torch.int32)
        if Ei is not None:
            am = torch.zeros(1, device=dE.device, dtype=torch.float32) if V == 0 else None   # (views add to dE afterwards)
            dWa, dWb, dba, dbb, dwc, dbc = attnpool_bwd_split_raw(Ei, Wa, Wb, wc, act_a, act_b, ds, dE, p_drop, seed, keep_a, keep_b,
                                                                  scores, m, l, d_main, row_bag, N if not ragged else 0, acc_e, am)
            if tok_after:
                d_tok = d_tok.float().contiguous()
                dti = split_image(d_tok, pad_rows=32)
                gate, chunk_max = split_tile_absmax(d_tok, chunks=True)   # one pass: 256-row tiles (dX) and 32-row chunks (dW)
                split_gemm_nt(dti, weight_image(Wtok.t().contiguous()), out=dE, accumulate=True, absmax_out=am, name="linear_bwd",
                              row_gate=gate, terms=GRAD_TERMS)
                dWtok = split_gemm_tn(Ei, dti, name="linear_bwd", b_chunk_max=chunk_max, terms=GRAD_TERMS)
                dbtok = d_tok.sum(0) if has_btok else None
            if am is not None:
                _put_absmax(dE, am)
        else:
            dWa, dWb, dba, dbb, dwc, dbc = attnpool_bwd_raw(E2d, Wa, Wb, wc, act_a, act_b, ds, dE, p_drop, seed, keep_a, keep_b,
                                                            scores, m, l, d_main, row_bag, N if not ragged else 0, acc_e)
        for i in range(V):   # ... and their dE terms are added once dE has been written (no read of E)
            vp, vm, vl = vflat[3 * i:3 * i + 3]
            pool_view_bwd_raw(E2d, scores, vp, vm, vl, d_pooled[:, 1 + i].contiguous(), dE, None, n_bags, N, views[i])
        return (dE.view(e_shape), dWa, dba, dWb, dbb, dwc, dbc, None, None, None, None, None, None, dWtok, dbtok, None, None, None) + (None,) * V


def attn_pool(E, Wa, ba, Wb, bb, wc, bc, p_drop=0.0, seed=0, keep_a=None, keep_b=None, cu_seqlens=None, max_len=None, views=(),
              tok_proj=None, e_img=None, e_only_image=False):
    """-> (pooled, raw scores), or (pooled, raw scores, token projections [T,P]) with tok_proj = (Wtok [P,H*512], btok or None).
    e_img = (image data, scale) of E when the producing kernel wrote one (split GEMM mode); e_only_image: E is that image tensor itself
    (no fp32 E was written) and the node's E-gradient is the image tensor's."""
    Wtok, btok = tok_proj if tok_proj is not None else (None, None)
    Eimg, Escale = e_img if e_img is not None else (None, None)
    pooled, scores, tok = AttnPoolFn.apply(E, Wa, ba, Wb, bb, wc, bc, float(p_drop), int(seed), keep_a, keep_b, cu_seqlens, max_len,
                                           None if Wtok is None else Wtok.contiguous(), None if btok is None else btok.contiguous(),
                                           Eimg, Escale, bool(e_only_image), *views)
    return (pooled, scores) if tok_proj is None else (pooled, scores, tok)


def gate_scores(E2d, Wa, ba, Wb, bb, wc, bc, p_drop=0.0, seed=0, keep_a=None, keep_b=None):
    return GateScoresFn.apply(E2d, Wa, ba, Wb, bb, wc, bc, float(p_drop), int(seed), keep_a, keep_b)


def softmax_pool(E, scores, cu_seqlens=None, max_len=None):
    return SoftmaxPoolFn.apply(E, scores, cu_seqlens, max_len)


# --------------------------------------------------------------------------------------------------
# N1: fused LayerNorm -> GELU -> Dropout (pre-attention MLP)
# --------------------------------------------------------------------------------------------------
class LNGeluDropFn(torch.autograd.Function):
    """y = Dropout_p(GELU(LayerNorm(x + bias; gamma, beta, eps)))  over the last axis (Model.py:351-354).
    `bias` (optional) is the bias of the preceding Linear: the GEMM then runs bias-free and the bias gradient (column
    sums of dx) comes out of the same backward pass instead of a separate reduction over dx."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, p_drop, seed, keep, bias):
        _require_act(x, "x")
        _require(gamma, "gamma")
        _require(beta, "beta")
        if bias is not None:
            _require(bias, "bias")
        lib = _native.lib()
        W = x.shape[-1]
        rows = x.numel() // W
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        with _timed("ln_gelu_drop_fwd", ("byte", 2.0 * x.numel() * x.element_size())):
            rc = getattr(lib, "mdl_ln_gelu_drop_fwd" + _sfx(x))(_ptr(x), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean),
                                                              _ptr(rstd), rows, W, float(eps), float(p_drop), int(seed),
                                                              _ptr(keep), _stream())
        if rc == -3:
            raise NotImplementedError("fused LayerNorm-GELU-Dropout supports widths 256/512/1024/2048/4096 (got %d)" % W)
        _native.check(rc, "mdl_ln_gelu_drop_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd, bias if bias is not None else torch.empty(0))
        ctx.cfg = (float(p_drop), int(seed), keep, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd, bias = ctx.saved_tensors
        p_drop, seed, keep, has_bias = ctx.cfg
        bias = bias if has_bias else None
        lib = _native.lib()
        W = x.shape[-1]
        rows = x.numel() // W
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        dbias = torch.empty_like(bias) if has_bias else None
        ws = _ws(lib.mdl_ln_gelu_drop_bwd_ws_bytes(rows, W), x.device)
        with _timed("ln_gelu_drop_bwd", ("byte", 3.0 * x.numel() * x.element_size())):
            rc = getattr(lib, "mdl_ln_gelu_drop_bwd" + _sfx(x))(_ptr(x), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(mean),
                                                              _ptr(rstd), _ptr(dy), _ptr(dx), _ptr(dg), _ptr(db), _ptr(dbias),
                                                              rows, W, p_drop, seed, _ptr(keep), _ptr(ws), _stream())
        _native.check(rc, "mdl_ln_gelu_drop_bwd")
        return dx, dg, db, None, None, None, None, dbias


class LNGeluDropGroupsFn(torch.autograd.Function):
    """LNGeluDropFn with one bias row per GROUP of rows (rows [cu_groups[g], cu_groups[g + 1]) take group_bias[g]): the first block of
    the pre-attention MLP when the stain encoding is folded out of its Linear (Model.py:125-132, :351) on the bf16 / exact-fp32 engines --
    the split engine adds the row in its GEMM epilogue instead (PreAttnBlockFn).  The backward returns the per-group sums of dx."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, p_drop, seed, keep, group_bias, cu_groups):
        _require_act(x, "x")
        _require(gamma, "gamma")
        _require(beta, "beta")
        _require(group_bias, "group_bias")
        _require(cu_groups, "cu_groups", torch.int64)
        lib = _native.lib()
        W = x.shape[-1]
        rows, G = x.numel() // W, group_bias.shape[0]
        if group_bias.shape != (G, W) or cu_groups.numel() != G + 1:
            raise ValueError("group_bias must be [G, W] with cu_groups [G + 1]")
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        with _timed("ln_gelu_drop_fwd", ("byte", 2.0 * x.numel() * x.element_size())):
            rc = getattr(lib, "mdl_ln_gelu_drop_fwd_groups" + _sfx(x))(_ptr(x), _ptr(group_bias), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean),
                                                                     _ptr(rstd), rows, W, float(eps), float(p_drop), int(seed), _ptr(keep),
                                                                     _ptr(cu_groups), G, _stream())
        if rc == -3:
            raise NotImplementedError("grouped LayerNorm-GELU-Dropout supports widths 256/512/1024 (got %d)" % W)
        _native.check(rc, "mdl_ln_gelu_drop_fwd_groups")
        ctx.save_for_backward(x, gamma, beta, mean, rstd, group_bias, cu_groups)
        ctx.cfg = (float(p_drop), int(seed), keep)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd, group_bias, cu_groups = ctx.saved_tensors
        p_drop, seed, keep = ctx.cfg
        lib = _native.lib()
        W = x.shape[-1]
        rows, G = x.numel() // W, group_bias.shape[0]
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dg, db, dgb = torch.empty_like(gamma), torch.empty_like(beta), torch.empty_like(group_bias)
        ws = _ws(lib.mdl_ln_gelu_drop_bwd_groups_ws_bytes(rows, W, G), x.device)
        with _timed("ln_gelu_drop_bwd", ("byte", 3.0 * x.numel() * x.element_size())):
            rc = getattr(lib, "mdl_ln_gelu_drop_bwd_groups" + _sfx(x))(_ptr(x), _ptr(group_bias), _ptr(gamma), _ptr(beta), _ptr(mean),
                                                                     _ptr(rstd), _ptr(dy), _ptr(dx), _ptr(dg), _ptr(db), _ptr(dgb), rows, W,
                                                                     p_drop, seed, _ptr(keep), _ptr(cu_groups), G, _ptr(ws), _stream())
        _native.check(rc, "mdl_ln_gelu_drop_bwd_groups")
        return dx, dg, db, None, None, None, None, dgb, None


def ln_gelu_drop_groups(x, gamma, beta, eps, p_drop, seed, keep, group_bias, cu_groups):
    return LNGeluDropGroupsFn.apply(x.contiguous(), gamma.contiguous(), beta.contiguous(), eps, p_drop, seed, keep,
                                    group_bias.float().contiguous(), cu_groups)


def ln_gelu_drop(x, gamma, beta, eps=1e-5, p_drop=0.0, seed=0, keep=None, bias=None):
    return LNGeluDropFn.apply(x.contiguous(), gamma.contiguous(), beta.contiguous(), eps, p_drop, seed, keep,
                              None if bias is None else bias.contiguous())


# --------------------------------------------------------------------------------------------------
# N1: bias-free Linear of the pre-attention MLP on the fp32 matrix cores
# --------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """Y = X W^T (+ bias) (Model.py:351, :355, :359 without the bias, which ln_gelu_drop adds; Model.py:140 token_projector and
    Model.py:145 projector with their bias); X [T,K] fp32 or bf16 (bf16 mode: x, y and their gradients bf16; W, bias and their
    gradients fp32 -- mdl_linear_*_bf16), W [N,K]."""

    @staticmethod
    def forward(ctx, x, W, bias):
        _require_act(x, "x")
        _require(W, "weight")
        if bias is not None:
            _require(bias, "bias")
        y = linear_fwd_raw(x, W, bias)
        ctx.save_for_backward(x, W)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW, db = linear_bwd_raw(x, W, dy.to(x.dtype).contiguous(), dx, ctx.has_bias)
        return dx, dW, db


LinearBf16Fn = LinearFn   # one node for both storage types (kept for callers of the round-2 name)


def linear_supported(x, W) -> bool:
    """Geometries of mdl_linear_* (include/madeleine_amd.h).  fp32: at most 256 rows -> K % 4 == 0, N % 4 == 0; otherwise
    N % 256 == 0 with K % 32 == 0, or N % 256 == 128 with K % 256 == 0.  bf16 activations (fp32 weight): N % 128 == 0,
    K % 32 == 0 (mdl_linear_*_bf16; at most 256 rows run the fp32 small-M kernel on an fp32 copy of the rows)."""
    N, K = W.shape
    if W.dtype != torch.float32 or not x.is_cuda:
        return False
    T = x.numel() // max(1, x.shape[-1])
    if x.dtype not in ACT_DTYPES:
        return False
    if T <= 256:
        return N % 4 == 0 and K % 4 == 0
    if x.dtype == torch.bfloat16:
        return bool(_native.lib().mdl_linear_bf16_supported(N, K, 1))
    return (N % 256 == 0 and K % 32 == 0) or (N % 128 == 0 and K % 256 == 0)


def linear(x, W, bias=None):
    """Linear over the last axis through the HIP kernels (fp32, or bf16 activations with fp32 parameters).  There is no library
    fallback: a geometry outside linear_supported raises (every Linear of the encoder -- reference Model.py:140, :145, :351, :355,
    :359 with patch_embedding_dim (+ 32 stain channels) a multiple of 32 -- is inside it)."""
    if not linear_supported(x, W):
        raise NotImplementedError(
            "madeleine_amd.linear: unsupported geometry / dtype (x %s %s on %s, weight %s %s); the HIP kernels need a ROCm device, "
            "fp32 weights, fp32 / bf16 activations and in_features a multiple of 32 (4 for at most 256 rows), out_features a "
            "multiple of 256 (or of 128 with in_features a multiple of 256; of 4 for at most 256 rows)"
            % (tuple(x.shape), x.dtype, x.device, tuple(W.shape), W.dtype))
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    Wc, bc = W.contiguous(), None if bias is None else bias.contiguous()
    if x.dtype == torch.bfloat16 and x2.shape[0] <= 256:
        # a handful of rows (one short bag under autocast): exact fp32 FMA kernel on the widened rows, result stored as bf16
        y = LinearFn.apply(x2.float(), Wc, bc).to(torch.bfloat16)
    elif x.dtype == torch.float32 and GEMM_MODE == "split" and split_linear_supported(x2.shape[0], W.shape[0], W.shape[1]):
        y = SplitLinearFn.apply(x2, Wc, bc)
    else:
        y = (LinearBf16Fn if x.dtype == torch.bfloat16 else LinearFn).apply(x2, Wc, bc)
    return y.view(*lead, W.shape[0])


# --------------------------------------------------------------------------------------------------
# L1: InfoNCE (batched over problems)
# --------------------------------------------------------------------------------------------------
class InfoNCEFn(torch.autograd.Function):
    """(loss [S], row_loss [S,Kmax]) for S padded problems Q,P [S,Kmax,D] with cnt[S] live rows each (loss.py:111-127);
    row_loss = the per-sample losses (reduction 'none').  Use ONE of the two outputs downstream."""

    @staticmethod
    def forward(ctx, Q, P, cnt, temperature, symmetric, per_row):
        _require(Q, "query")
        _require(P, "positive_key")
        _require(cnt, "cnt", torch.int32)
        lib = _native.lib()
        S, Kmax, D = Q.shape
        loss = torch.empty(S, device=Q.device, dtype=torch.float32)
        rows = torch.empty(S, Kmax, device=Q.device, dtype=torch.float32) if per_row else None
        ws = _ws(lib.mdl_infonce_ws_bytes(S, Kmax, D), Q.device)
        with _timed("infonce_fwd"):
            rc = lib.mdl_infonce_fwd(_ptr(Q), _ptr(P), _ptr(cnt), _ptr(loss), _ptr(rows), S, Kmax, D, float(temperature),
                                     int(symmetric), _ptr(ws), _stream())
        _native.check(rc, "mdl_infonce_fwd")
        ctx.save_for_backward(cnt, ws, Q, P)
        ctx.cfg = (S, Kmax, D, float(temperature), int(symmetric), bool(per_row))
        if not per_row:
            rows = loss.new_empty(0)
            ctx.mark_non_differentiable(rows)
        return loss, rows

    @staticmethod
    def backward(ctx, d_loss, d_rows):
        cnt, ws, Q, P = ctx.saved_tensors
        S, Kmax, D, temperature, symmetric, per_row = ctx.cfg
        lib = _native.lib()
        dev = cnt.device
        dQ = torch.empty(S, Kmax, D, device=dev, dtype=torch.float32)
        dP = torch.empty_like(dQ)
        if per_row:
            # loss [S] = mean of the rows: fold an upstream gradient on it into the per-row gradients
            g = d_rows.float().contiguous() if d_rows is not None else torch.zeros(S, Kmax, device=dev)
            if d_loss is not None:
                g = g + d_loss.float().unsqueeze(1) / cnt.clamp_min(1).unsqueeze(1).float()
            g = g.contiguous()
            with _timed("infonce_bwd"):
                rc = lib.mdl_infonce_bwd(_ptr(Q), _ptr(P), None, _ptr(g), _ptr(cnt), _ptr(dQ), _ptr(dP), S, Kmax, D, temperature,
                                         symmetric, _ptr(ws), _stream())
        else:
            d_loss = d_loss.float().contiguous()
            with _timed("infonce_bwd"):
                rc = lib.mdl_infonce_bwd(_ptr(Q), _ptr(P), _ptr(d_loss), None, _ptr(cnt), _ptr(dQ), _ptr(dP), S, Kmax, D, temperature,
                                         symmetric, _ptr(ws), _stream())
        _native.check(rc, "mdl_infonce_bwd")
        return dQ, dP, None, None, None, None


def info_nce_batched(Q, P, cnt, temperature, symmetric):
    return InfoNCEFn.apply(Q, P, cnt, float(temperature), bool(symmetric), False)[0]


def info_nce_rows(Q, P, cnt, temperature, symmetric):
    """Per-sample losses [S, Kmax] (reduction 'none'); rows >= cnt[s] are zero."""
    return InfoNCEFn.apply(Q, P, cnt, float(temperature), bool(symmetric), True)[1]


# --------------------------------------------------------------------------------------------------
# G0-G3: graph optimal transport
# --------------------------------------------------------------------------------------------------
class GOTFn(torch.autograd.Function):
    """out[2] = (sum_b WD_b, sum_b GWD_b) for token sets V,Q [k,n,d] (loss.py:278-302 after the sub-sampling).

    minmax_in (optional float[6] device tensor) replaces the batch-local threshold extrema (data-parallel path);
    the second output is this call's own extrema [6] (non-differentiable)."""

    @staticmethod
    def forward(ctx, V, Q, minmax_in, reduce_dminmax):
        _require(V, "v_")
        _require(Q, "q_")
        if V.shape != Q.shape or V.dim() != 3:
            raise ValueError("GOT expects two token tensors of identical shape [k, n, d]")
        lib = _native.lib()
        k, n, d = V.shape
        nbytes = lib.mdl_got_ws_bytes(k, n, d)
        if nbytes == -3:
            raise NotImplementedError("madeleine_amd.GOT supports n <= 512 tokens per bag and d <= 128 (got n=%d, d=%d); "
                                      "the reference calls it with subsample=256 (trainer.py:44)" % (n, d))
        ws = _ws(nbytes, V.device)
        out = torch.empty(2, device=V.device, dtype=torch.float32)
        mm = torch.empty(6, device=V.device, dtype=torch.float32)
        with _timed("got_fwd"):
            rc = lib.mdl_got_fwd(_ptr(V), _ptr(Q), _ptr(out), _ptr(mm), _ptr(minmax_in), k, n, d, _ptr(ws), _stream())
        _native.check(rc, "mdl_got_fwd")
        ctx.save_for_backward(V, Q, ws)
        ctx.reduce_dminmax = reduce_dminmax
        ctx.mark_non_differentiable(mm)
        return out, mm

    @staticmethod
    def backward(ctx, d_out, _d_mm):
        V, Q, ws = ctx.saved_tensors
        lib = _native.lib()
        k, n, d = V.shape
        dV, dQ = torch.empty_like(V), torch.empty_like(Q)
        d_out = d_out.contiguous()
        if ctx.reduce_dminmax is None:
            with _timed("got_bwd"):
                rc = lib.mdl_got_bwd(_ptr(V), _ptr(Q), _ptr(d_out), _ptr(dV), _ptr(dQ), k, n, d, _ptr(ws), _stream())
            _native.check(rc, "mdl_got_bwd")
        else:
            dmm = torch.empty(6, device=V.device, dtype=torch.float32)
            rc = lib.mdl_got_bwd_begin(_ptr(d_out), _ptr(dmm), k, n, d, _ptr(ws), _stream())
            _native.check(rc, "mdl_got_bwd_begin")
            dmm = ctx.reduce_dminmax(dmm).contiguous()      # e.g. all_reduce(SUM) over ranks
            rc = lib.mdl_got_bwd_finish(_ptr(V), _ptr(Q), _ptr(dV), _ptr(dQ), _ptr(dmm), k, n, d, _ptr(ws), _stream())
            _native.check(rc, "mdl_got_bwd_finish")
        return dV, dQ, None, None


def got_extrema(V, Q):
    """Extrema [6] (cross min,max | intra-V min,max | intra-Q min,max) of this batch's raw GOT cost tensors."""
    _require(V, "v_")
    _require(Q, "q_")
    lib = _native.lib()
    k, n, d = V.shape
    ws = _ws(lib.mdl_got_ws_bytes(k, n, d), V.device)
    mm = torch.empty(6, device=V.device, dtype=torch.float32)
    rc = lib.mdl_got_extrema(_ptr(V), _ptr(Q), _ptr(mm), k, n, d, _ptr(ws), _stream())
    _native.check(rc, "mdl_got_extrema")
    return mm


def got_exchange_timeouts(ws) -> float:
    """Diagnostic of the split IPOT sweeps (csrc/got_impl.inc, Xch): non-zero when a workgroup gave up waiting for its partner's column
    sums in the last pass on this workspace.  That pass's numbers are void and say so themselves: the distances of a forward and the
    token gradients of a backward on a flagged workspace come back NaN (got_sum_kernel / got_cost_bwd_kernel) -- the flag is for
    diagnosis, nothing has to poll it.  The workspace's global region ends with
    {generation, time-out flag, 0, 0}; mdl_got_ws_bytes adds 64 bytes of padding behind it.  Synchronises."""
    nf = (ws.numel() * ws.element_size() - 64) // 4
    return float(ws.view(torch.uint8)[:nf * 4].view(torch.float32)[nf - 3])


def _got_set_exchange_timeout(ws, value: float = 1.0) -> None:
    """Test hook: writes the workspace's time-out flag (what a split sweep does when it gives up on its partner)."""
    nf = (ws.numel() * ws.element_size() - 64) // 4
    ws.view(torch.uint8)[:nf * 4].view(torch.float32)[nf - 3] = value


def got(V, Q, minmax_in=None, reduce_dminmax=None, return_extrema=False):
    out, mm = GOTFn.apply(V, Q, minmax_in, reduce_dminmax)
    return (out, mm) if return_extrema else out


class HipGotImpl:
    """The four stages of GOT on the C ABI, as used by the multi-problem / data-parallel autograd node
    (madeleine_amd.distributed.got_multi).  A CPU implementation with the same interface exists only in tests/."""

    @staticmethod
    def extrema(V, Q):
        return got_extrema(V, Q)

    @staticmethod
    def forward(V, Q, minmax):
        lib = _native.lib()
        k, n, d = V.shape
        nbytes = lib.mdl_got_ws_bytes(k, n, d)
        if nbytes == -3:
            raise NotImplementedError("madeleine_amd.GOT supports n <= 512 tokens per bag and d <= 128 (got n=%d, d=%d)" % (n, d))
        ws = _ws(nbytes, V.device)
        out = torch.empty(2, device=V.device, dtype=torch.float32)
        mm = minmax.contiguous()
        with _timed("got_fwd"):
            rc = lib.mdl_got_fwd(_ptr(V), _ptr(Q), _ptr(out), None, _ptr(mm), k, n, d, _ptr(ws), _stream())
        _native.check(rc, "mdl_got_fwd")
        return out, (V, Q, ws)

    @staticmethod
    def backward_begin(state, d_out):
        V, Q, ws = state
        lib = _native.lib()
        k, n, d = V.shape
        dmm = torch.empty(6, device=V.device, dtype=torch.float32)
        with _timed("got_bwd"):
            rc = lib.mdl_got_bwd_begin(_ptr(d_out.contiguous()), _ptr(dmm), k, n, d, _ptr(ws), _stream())
        _native.check(rc, "mdl_got_bwd_begin")
        return dmm

    @staticmethod
    def backward_finish(state, dmm_total):
        V, Q, ws = state
        lib = _native.lib()
        k, n, d = V.shape
        dV, dQ = torch.empty_like(V), torch.empty_like(Q)
        dmm_total = dmm_total.contiguous()
        with _timed("got_bwd_finish"):
            rc = lib.mdl_got_bwd_finish(_ptr(V), _ptr(Q), _ptr(dV), _ptr(dQ), _ptr(dmm_total), k, n, d, _ptr(ws), _stream())
        _native.check(rc, "mdl_got_bwd_finish")
        return dV, dQ

    # ---- several problems per launch (mdl_got_*_multi): every kernel launch covers all problems, on the caller's stream alone ----
    MAX_BATCH = 4

    @staticmethod
    def can_batch(problems) -> bool:
        """2 .. MAX_BATCH non-empty problems on one ROCm device, every n <= 256, one d."""
        if not 2 <= len(problems) <= HipGotImpl.MAX_BATCH or os.environ.get("MADELEINE_GOT_NO_BATCH"):
            return False
        d = problems[0][0].shape[2]
        return all(V.is_cuda and V.dtype == torch.float32 and V.shape[0] >= 1 and 1 <= V.shape[1] <= 256 and V.shape[2] == d
                   and V.device == problems[0][0].device for V, _ in problems)

    @staticmethod
    def _arrays(problems, wss):
        import ctypes
        np_ = len(problems)
        ptrs = lambda ts: (ctypes.c_void_p * np_)(*[t.data_ptr() for t in ts])      # noqa: E731
        ks = (ctypes.c_int * np_)(*[int(V.shape[0]) for V, _ in problems])
        ns = (ctypes.c_int * np_)(*[int(V.shape[1]) for V, _ in problems])
        return np_, ptrs([V for V, _ in problems]), ptrs([Q for _, Q in problems]), ks, ns, int(problems[0][0].shape[2]), ptrs(wss), ptrs

    @staticmethod
    def _rows(t):
        import ctypes
        return (ctypes.c_void_p * t.shape[0])(*[t.data_ptr() + i * t.stride(0) * t.element_size() for i in range(t.shape[0])])

    @staticmethod
    def extrema_multi(problems):
        lib = _native.lib()
        dev = problems[0][0].device
        wss = [_ws(lib.mdl_got_ws_bytes(*V.shape), dev) for V, _ in problems]
        np_, Vp, Qp, ks, ns, d, wsp, _ = HipGotImpl._arrays(problems, wss)
        mm = torch.empty(np_, 6, device=dev, dtype=torch.float32)
        rc = lib.mdl_got_extrema_multi(np_, Vp, Qp, HipGotImpl._rows(mm), ks, ns, d, wsp, _stream())
        _native.check(rc, "mdl_got_extrema_multi")
        return mm

    @staticmethod
    def forward_multi(problems, minmax):
        lib = _native.lib()
        dev = problems[0][0].device
        for V, _ in problems:
            if lib.mdl_got_ws_bytes(*V.shape) == -3:
                raise NotImplementedError("madeleine_amd.GOT supports n <= 512 tokens per bag and d <= 128")
        wss = [_ws(lib.mdl_got_ws_bytes(*V.shape), dev) for V, _ in problems]
        np_, Vp, Qp, ks, ns, d, wsp, _ = HipGotImpl._arrays(problems, wss)
        out = torch.empty(np_, 2, device=dev, dtype=torch.float32)
        mm = minmax.contiguous()
        with _timed("got_fwd"):
            rc = lib.mdl_got_fwd_multi(np_, Vp, Qp, HipGotImpl._rows(out), HipGotImpl._rows(mm), ks, ns, d, wsp, _stream())
        _native.check(rc, "mdl_got_fwd_multi")
        return out, (list(problems), wss)

    @staticmethod
    def backward_begin_multi(state, d_outs):
        problems, wss = state
        lib = _native.lib()
        np_, _Vp, _Qp, ks, ns, d, wsp, _ = HipGotImpl._arrays(problems, wss)
        d_outs = d_outs.contiguous()
        dmm = torch.empty(np_, 6, device=d_outs.device, dtype=torch.float32)
        with _timed("got_bwd"):
            rc = lib.mdl_got_bwd_begin_multi(np_, HipGotImpl._rows(d_outs), HipGotImpl._rows(dmm), ks, ns, d, wsp, _stream())
        _native.check(rc, "mdl_got_bwd_begin_multi")
        return dmm

    @staticmethod
    def backward_finish_multi(state, dmm_total):
        problems, wss = state
        lib = _native.lib()
        np_, Vp, Qp, ks, ns, d, wsp, ptrs = HipGotImpl._arrays(problems, wss)
        dVs, dQs = [torch.empty_like(V) for V, _ in problems], [torch.empty_like(Q) for _, Q in problems]
        dmm_total = dmm_total.contiguous()
        with _timed("got_bwd_finish"):
            rc = lib.mdl_got_bwd_finish_multi(np_, Vp, Qp, ptrs(dVs), ptrs(dQs), HipGotImpl._rows(dmm_total), ks, ns, d, wsp, _stream())
        _native.check(rc, "mdl_got_bwd_finish_multi")
        return list(zip(dVs, dQs))
