#!/usr/bin/env python3
"""bf16 gate forward alone at config-2 geometry, for rocprofv3 passes.  --tile 128|256, --dm (dropout on), --iters."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tile", default="256")
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--p", type=float, default=0.25)
ap.add_argument("--split", action="store_true", help="the split-fp16 engine's gate forward (fp32 E image) instead of the bf16 one")
a = ap.parse_args()
if a.tile == "128":
    os.environ["MADELEINE_BF16_GATE128"] = "1"
dev = torch.device("cuda:0")
T, H = 262144, 4
g = torch.Generator(device=dev).manual_seed(0)
E = torch.randn(T, H * 512, device=dev, generator=g)
s = 1 / 512 ** 0.5
Wa = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
Wb = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
ba, bb, wc = [(torch.rand(H, 512, device=dev, generator=g) * 2 - 1) * s for _ in range(3)]
bc = torch.zeros(H, device=dev)
if a.split:
    Ei = MF.split_image(E)
    for it in range(a.iters):
        MF.gate_fwd_split_raw(Ei, Wa, ba, Wb, bb, wc, bc, a.p, 7 + it, None, None, True)
else:
    Eb = E.to(torch.bfloat16)
    for it in range(a.iters):
        MF.gate_fwd_raw(Eb, Wa, ba, Wb, bb, wc, bc, a.p, 7 + it, None, None, True)
torch.cuda.synchronize()
print("done")
