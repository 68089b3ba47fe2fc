#!/usr/bin/env python3
"""Times GOT forward+backward (one stain) at a few (k, n) geometries: python tools/bench_got.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
import json
GEOMS = json.loads(os.environ.get('GOT_GEOMS', '[[32,32],[32,64],[32,128],[32,192],[32,256]]'))
for (k, n) in GEOMS:
    g = torch.Generator(device=dev).manual_seed(0)
    v = torch.randn(k, n, 128, device=dev, generator=g).requires_grad_()
    q = (torch.randn(k, n, 128, device=dev, generator=g) + 0.7 * v.detach()).requires_grad_()
    ts = []
    for it in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); o = MF.got(v, q); e[1].record(); (o[0] + o[1]).backward(); e[2].record(); torch.cuda.synchronize()
        ts.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    f, b = ts[-1]
    print(f"k={k} n={n}: fwd {f:8.3f} ms  bwd {b:8.3f} ms  value {float(o.sum()):.5f}")
