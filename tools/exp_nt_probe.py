"""Where a tile's time goes in the split NT product: s_memtime stamps of every workgroup (variant library built with -DMDL_SP_PROBE:
entry | main loop start | main loop end | exit | hardware id), grouped by compute unit: set-up, main loop, epilogue, and the gap between a
workgroup's exit and the next workgroup's entry on the same unit.  Run with MADELEINE_LIB=tools/ab/probe.so."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madeleine_amd import _native
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T = 262144
lib = ctypes.CDLL(os.environ.get("MADELEINE_LIB") or _native.lib_path())
lib.mdl_debug_sp_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
torch.manual_seed(0)
for (N, K) in [(512, 512), (2048, 512), (512, 2048)]:
    a = torch.randn(T, K, device=dev)
    b = 0.05 * torch.randn(N, K, device=dev)
    A, B = MF.split_image(a), MF.weight_image(b)
    for _ in range(3):
        MF.split_gemm_nt(A, B)
    torch.cuda.synchronize()
    n_wg = min(8192, (T // 256) * (N // 256))
    buf = np.zeros((n_wg, 5), dtype=np.uint64)
    rc = lib.mdl_debug_sp_probe_read(buf.ctypes.data, n_wg)
    assert rc == 0, rc
    t = buf[:, :4].astype(np.int64)
    hw = buf[:, 4]
    cu = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64) * 4096 + (hw & np.uint64(0xFFFFFFFF)).astype(np.int64) // 256 % 4096   # (xcc, se/cu bits of HW_ID)
    setup, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        order = idx[np.argsort(t[idx, 0])]
        gaps += list(t[order[1:], 0] - t[order[:-1], 3])
    gaps = np.array(gaps)
    span = t[:, 3].max() - t[:, 0].min()
    print("N %4d K %4d: %d workgroups on %d units; ticks (median): set-up %d, main loop %d (%d per chunk), epilogue %d, exit->next entry on the unit %d "
          "(p10 %d p90 %d); kernel span %d ticks = %.1f rounds x %d" % (N, K, n_wg, len(np.unique(cu)), np.median(setup), np.median(loop), np.median(loop) // (K // 32),
          np.median(epi), np.median(gaps) if len(gaps) else -1, np.percentile(gaps, 10) if len(gaps) else -1, np.percentile(gaps, 90) if len(gaps) else -1, span,
          n_wg / max(1, len(np.unique(cu))), np.median(t[:, 3] - t[:, 0])), flush=True)
