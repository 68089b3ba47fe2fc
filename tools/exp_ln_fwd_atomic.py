#!/usr/bin/env python3
"""mdl_ln_gelu_drop_fwd_split at 512 / 2048 columns with and without the rstd_max accumulator (one atomicMax per wave at the end)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import _native
from madeleine_amd import functional as MF
lib = _native.lib()
dev = torch.device("cuda:0")
P = MF._ptr
for T in (30000, 262144):
    for W in (512, 2048):
        x = torch.randn(T, W, device=dev)
        g, b, lb = torch.ones(W, device=dev), torch.zeros(W, device=dev), torch.zeros(W, device=dev)
        img = torch.empty(T, W * 4, device=dev, dtype=torch.uint8)
        sc = torch.zeros(2, device=dev); mean = torch.empty(T, device=dev); rstd = torch.empty(T, device=dev)
        rmax = torch.zeros(1, device=dev)
        for with_max in (False, True):
            def call():
                rc = lib.mdl_ln_gelu_drop_fwd_split(P(x), P(lb), P(g), P(b), None, P(img), P(sc), P(mean), P(rstd), T, W, 1e-5, 0.1, 1234, None,
                                                    None, P(rmax) if with_max else None, MF._stream())
                assert rc == 0, rc
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print("T %6d W %4d rstd_max %-5s: %.1f us  %.2f TB/s" % (T, W, with_max, ms * 1e3, T * W * 8 / ms / 1e9))
