#!/usr/bin/env python3
"""Runs the hot-path kernels alone at BASELINE config-2 geometry (64 bags x 4096 tokens x 2048 ch) so that
rocprofv3 passes (--kernel-trace --stats, or one --pmc set per pass) stay short.  Usage:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python tools/prof_kernels.py [--iters 5] [--only pool|gate]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", default="all")
ap.add_argument("--bags", type=int, default=64)
ap.add_argument("--tokens", type=int, default=4096)
ap.add_argument("--image", action="store_true", help="--only pool: E as a split image (what the split GEMM mode's step pools from)")
a = ap.parse_args()
dev = torch.device("cuda:0")
BM, N, H = a.bags, a.tokens, 4
g = torch.Generator(device=dev).manual_seed(0)
E = torch.randn(BM, N, H * 512, device=dev, generator=g)
s = 1.0 / 512 ** 0.5
Wa = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
Wb = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
ba, bb, wc = [(torch.rand(H, 512, device=dev, generator=g) * 2 - 1) * s for _ in range(3)]
bc = torch.zeros(H, device=dev)
E2 = E.view(BM * N, -1)
dpool = torch.randn(BM, H * 512, device=dev, generator=g)
for it in range(a.iters):
    if a.only in ("all", "gate"):
        scores, aa, ab = MF.gate_fwd_raw(E2, Wa, ba, Wb, bb, wc, bc, 0.25, 123 + it, None, None, True)
    else:
        scores = torch.randn(BM * N, H, device=dev, generator=g)
    if a.only == "pool" and a.image:
        Ei = MF.split_image(E2)
        pooled, m, l = MF.pool_fwd_img_raw(Ei, scores, BM, N, None, N)
        ds = torch.empty_like(scores)
        MF.pool_dscores_img_raw(Ei, scores, pooled, m, l, dpool, ds, 0, BM, N, None, N)
    elif a.only in ("all", "pool"):
        pooled, m, l = MF.pool_fwd_raw(E2, scores, BM, N, None, N)
        dE = torch.empty_like(E2)
        ds = torch.empty_like(scores)
        # product path (functional.AttnPoolFn.backward): scores-only pooling backward, its dE term is added by the gate dX
        MF.pool_bwd_raw(E2, scores, pooled, m, l, dpool, None if a.only == "all" else dE, 0, ds, 0, BM, N, None, N)
    if a.only in ("all", "gate"):
        if a.only == "gate":
            dE = torch.empty_like(E2)
            ds = torch.randn(BM * N, H, device=dev, generator=g)
            MF.gate_bwd_raw(E2, Wa, Wb, wc, aa, ab, ds, dE, 0, 0.25, 123 + it, None, None)
        else:
            MF.attnpool_bwd_raw(E2, Wa, Wb, wc, aa, ab, ds, dE, 0.25, 123 + it, None, None, scores, m, l, dpool, None, N)
torch.cuda.synchronize()
print("done", a.iters)
