#!/usr/bin/env python3
"""What does the split NT tile loop wait for?  Times mdl_split_gemm_nt (K = 2048 -> N = 512 and K = 512 -> N = 2048 at T = 262,144)
with the library given in MADELEINE_LIB: probe builds (tools/micro/loop_probes.patch applied to csrc/split_engine.hpp, then
tools/ab/build_variant.sh with -DMDL_SP_PROBE_NODMA / _NOBARRIER / _NOWAIT / -DMDL_SP_SETPRIO) drop one ingredient
of the loop each -- their results are WRONG, only their times mean something.  Usage: exp_loop_probes.py <label>"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T = 262144
label = sys.argv[1] if len(sys.argv) > 1 else "base"
res = []
for K, N in ((2048, 512), (512, 2048)):
    x = torch.randn(T, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    A, B = MF.split_image(x), MF.weight_image(w)
    out = torch.empty(T, N, device=dev)
    for _ in range(5):
        MF.split_gemm_nt(A, B, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        MF.split_gemm_nt(A, B, out=out)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20)
    del x, w, A, B, out
t_long, t_short = res   # 8 tiles x 64 chunks | 32 tiles x 16 chunks per CU
chunk = (t_long - t_short / 4) / (512 - 128) * 1e3          # 512 c + 8 o = t_long ; 512 c + 32 o = t_short
tile = (t_short - t_long) / 24 * 1e3
print("%-8s K2048->512 %.3f ms  K512->2048 %.3f ms  => %.2f us per 32-k chunk, %.1f us per tile" % (label, t_long, t_short,
      (t_long * 1e3 - 8 * tile) / 512, tile))
