#!/usr/bin/env python3
"""bf16 gate forward (config-2 geometry) with the co-resident workgroups' phases staggered by MADELEINE_GATE_STAGGER x 64 clocks
(gate_common.hpp:gate_stagger): does overlapping one workgroup's VALU epilogue with the other's MFMA main loop pay?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF  # noqa: E402

dev = torch.device("cuda:0")
T, H = 262144, 4
g = torch.Generator(device=dev).manual_seed(0)
E = torch.randn(T, H * 512, device=dev, generator=g).to(torch.bfloat16)
s = 1 / 512 ** 0.5
Wa = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
Wb = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
ba, bb, wc = [(torch.rand(H, 512, device=dev, generator=g) * 2 - 1) * s for _ in range(3)]
bc = torch.zeros(H, device=dev)


def run(n=8):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc, aa, ab = MF.gate_fwd_raw(E, Wa, ba, Wb, bb, wc, bc, 0.25, 7, None, None, True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], float(sc.double().sum())


flop = T * H * 2 * 512 * 1024
for tile in ("256", "128"):
    if tile == "128":
        os.environ["MADELEINE_BF16_GATE128"] = "1"
    else:
        os.environ.pop("MADELEINE_BF16_GATE128", None)
    for st in ([0] if tile == "256" else [0, 0, 127, 254, 30000, -127, -254, -508, -30000, 0]):
        os.environ["MADELEINE_GATE_STAGGER"] = str(st)
        ms, chk = run()
        print(f"tile {tile} stagger {st:4d} x64 clk: {ms:.3f} ms  {flop / ms / 1e9:.0f} TF   checksum {chk:.4f}", flush=True)
