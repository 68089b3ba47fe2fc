#!/usr/bin/env python3
"""Pooling forward / scores-backward alone at BASELINE config-2 geometry (64 bags x 4096 tokens x 2048 ch), timed with events over
many launches: A/B of kernel variants (MADELEINE_LIB=tools/ab/<name>.so).  Prints ms and the algorithmic HBM rate of each."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--bags", type=int, default=64)
ap.add_argument("--tokens", type=int, default=4096)
a = ap.parse_args()
dev = torch.device("cuda:0")
BM, N, H = a.bags, a.tokens, 4
g = torch.Generator(device=dev).manual_seed(0)
E2 = torch.randn(BM * N, H * 512, device=dev, generator=g)
scores = torch.randn(BM * N, H, device=dev, generator=g)
dpool = torch.randn(BM, H * 512, device=dev, generator=g)


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


for name, Ex, fwd, bwd in (
    ("image", MF.split_image(E2), MF.pool_fwd_img_raw, MF.pool_dscores_img_raw),
    ("fp32", E2, MF.pool_fwd_raw, None),
):
    nbytes = BM * N * H * 512 * 4
    out = fwd(Ex, scores, BM, N, None, N)
    ms = timed(lambda: fwd(Ex, scores, BM, N, None, N))
    print(f"pool_fwd[{name}] {ms:.4f} ms  {nbytes / ms / 1e9:.3f} TB/s  checksum {out[0].double().sum().item():.10e}")
    if bwd is not None:
        pooled, m, l = out
        ds = torch.empty_like(scores)
        ms = timed(lambda: bwd(Ex, scores, pooled, m, l, dpool, ds, 0, BM, N, None, N))
        print(f"pool_dscores[{name}] {ms:.4f} ms  {nbytes / ms / 1e9:.3f} TB/s  checksum {ds.double().sum().item():.10e}")
