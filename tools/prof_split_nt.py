#!/usr/bin/env python3
"""The split NT product alone on the Linear shapes of config 2 (for rocprofv3 --kernel-trace / --pmc): what the tile loop waits for.
Usage: prof_split_nt.py [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T = 262144
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for K, N in ((2048, 512), (512, 2048)):
    x = torch.randn(T, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    A, B = MF.split_image(x), MF.weight_image(w)
    out = torch.empty(T, N, device=dev)
    for _ in range(iters):
        MF.split_gemm_nt(A, B, out=out)
    torch.cuda.synchronize()
    del x, w, A, B, out
