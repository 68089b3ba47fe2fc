"""A/B of the forward C_gamma products by row halves (STAGE 4 of got_fwd_stage_kernel) against one workgroup per case, same process:
MADELEINE_GOT_NO_HALF_PRODUCTS toggled between calls.  Values, gradients, forward / backward times at rank-like shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madeleine_amd import _native
from tools.got_ab import load, run, rel

L = load(_native.lib_path())
dev = torch.device("cuda:0")
for (k, n) in [(128, 256), (32, 256), (94, 200), (128, 150), (6, 256)]:
    g = torch.Generator(device=dev).manual_seed(k * 1000 + n)
    v = torch.randn(k, n, 128, device=dev, generator=g)
    q = torch.randn(k, n, 128, device=dev, generator=g) + 0.7 * v
    ref = None
    for rep in range(2):
        for mode in ("halves", "whole"):
            if mode == "whole":
                os.environ["MADELEINE_GOT_NO_HALF_PRODUCTS"] = "1"
            else:
                os.environ.pop("MADELEINE_GOT_NO_HALF_PRODUCTS", None)
            o, dv, dq, t, err = run(L, v, q, reps=3)
            if ref is None:
                ref = (o, dv, dq)
            print(f"k={k:3d} n={n:3d} {mode:<7} wd {float(o[0]):.6f} gw {float(o[1]):.6f}  out rel {rel(o, ref[0]):.2e} dV rel {rel(dv, ref[1]):.2e} "
                  f"dQ rel {rel(dq, ref[2]):.2e}   {t[0]:7.3f} + {t[1]:7.3f} ms  flag {err}", flush=True)
os.environ.pop("MADELEINE_GOT_NO_HALF_PRODUCTS", None)
