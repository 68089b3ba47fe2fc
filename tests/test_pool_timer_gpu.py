"""Dispatch timer of the A3 forward (include/madeleine_amd.h: mdl_pool_timer_arm / _read; functional.PoolDispatchTimer) -- the clock
behind bench.py's roofline.achieved: arming it must not change a bit of the result (reference semantics: softmax over patches + weighted
sum, madeleine/models/Model.py:400-417), the three durations must be consistent, and only armed launches fill a slot."""
import pytest
import torch

from tests._util import t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_armed_launches_are_the_same_kernels_and_report_consistent_times(dev):
    from madeleine_amd import functional as MF
    B, N, H = 6, 3000, 4
    E = t((B * N, H * 512), "pt:E").to(dev)
    s = t((B * N, H), "pt:s").to(dev) * 3
    base = MF.pool_fwd_raw(E, s, B, N, None, N)
    MF.POOL_TIMER = MF.PoolDispatchTimer()
    try:
        outs = [MF.pool_fwd_raw(E, s, B, N, None, N) for _ in range(5)]
        rep = MF.POOL_TIMER.report()
    finally:
        MF.POOL_TIMER = None
    after = MF.pool_fwd_raw(E, s, B, N, None, N)          # disarmed again: plain launches
    for o in outs + [after]:
        for a, b in zip(o, base):
            assert torch.equal(a, b)
    assert len(rep) == 5
    for part, comb, span in rep:
        assert 0.0 < part < 5.0 and 0.0 < comb < 1.0
        assert span >= part and span >= comb and span < part + comb + 0.5   # the gap between the two dispatches is microseconds
    # algorithmic bytes / pool_partial time stays below the HBM peak (a zero or bogus timestamp would not)
    gbs = (B * N * H * (512 * 4 + 4)) / (min(r[0] for r in rep) * 1e-3) / 1e9
    assert 50.0 < gbs < 8000.0, gbs


def test_unused_slot_is_refused(dev):
    import ctypes
    from madeleine_amd import _native
    lib = _native.lib()
    ms = (ctypes.c_float * 3)()
    assert lib.mdl_pool_timer_arm(63) == 0          # armed, never launched
    assert lib.mdl_pool_timer_read(63, ms) != 0
    assert lib.mdl_pool_timer_arm(-1) == 0          # disarm
    assert lib.mdl_pool_timer_arm(64) != 0
