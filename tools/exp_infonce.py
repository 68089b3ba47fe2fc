"""InfoNCE forward + backward, fused (one launch each) (MADELEINE_INFONCE_FUSED=1) against the staged kernels, torch events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
for S, k in ((3, 32), (4, 64), (4, 256)):
    Q = torch.randn(S, k, 512, device=dev).requires_grad_()
    P = (torch.randn(S, k, 512, device=dev) + 0.2 * Q.detach()).requires_grad_()
    cnt = torch.full((S,), k, dtype=torch.int32, device=dev)
    for staged in (False, True):
        if not staged:
            os.environ["MADELEINE_INFONCE_FUSED"] = "1"
        else:
            os.environ.pop("MADELEINE_INFONCE_FUSED", None)
        def step():
            Q.grad = P.grad = None
            MF.info_nce_batched(Q, P, cnt, 0.001, True).sum().backward()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        MF.TIMER = MF.KernelTimer()
        for _ in range(20):
            step()
        r = MF.TIMER.report()
        MF.TIMER = None
        print("S=%d k=%d %s: fwd %.1f us  bwd %.1f us" % (S, k, "staged" if staged else "fused ", r["infonce_fwd"][0] * 1e3, r["infonce_bwd"][0] * 1e3))
