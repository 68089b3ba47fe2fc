#!/usr/bin/env python3
"""EXPERIMENT: do the 256 workgroups of a split-engine GEMM run their epilogues (256 KiB of stores each) in lockstep?  Times
mdl_split_gemm_nt on the three Linear shapes of config 2 with the first wave of workgroups delayed by phase * (tile period / P).
Needs tools/micro/tile_stagger.patch applied (the MADELEINE_SP_STAGGER hook is not in the product).  Result (round 4): alone, the
K = 512 -> N = 2048 product gains 7 % and the two shorter launches lose 3-9 % to the tail; inside the config-2 step every variant is
null (23.18-23.28 ms with 0 / 2 / 4 / 8 phases) -- the epilogue bursts are not what the tile loop waits for."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T = 262144
for K, N in ((512, 2048), (512, 512), (2048, 512)):
    x = torch.randn(T, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    A, B = MF.split_image(x), MF.weight_image(w)
    out = torch.empty(T, N, device=dev)
    for P in (0, 2, 4, 8, 16):
        if P:
            os.environ["MADELEINE_SP_STAGGER"] = str(P)
        else:
            os.environ.pop("MADELEINE_SP_STAGGER", None)
        for _ in range(3):
            MF.split_gemm_nt(A, B, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            MF.split_gemm_nt(A, B, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("K %4d N %4d  phases %2d : %.3f ms  %.0f TF fp32-equivalent" % (K, N, P, ms, 2.0 * T * N * K / ms / 1e9), flush=True)
    del x, w, A, B, out
