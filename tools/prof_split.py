"""The split-engine kernels alone at config-2 geometry (short rocprofv3 passes): gate fwd / dz / dX / dW and the 512 -> 2048 Linear."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import functional as MF

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--tokens", type=int, default=262144)
a = ap.parse_args()
dev = torch.device("cuda:0")
T, H = a.tokens, 4
g = torch.Generator(device=dev).manual_seed(1)
E = torch.randn(T, H * 512, device=dev, generator=g).requires_grad_()
s = 1 / 512 ** 0.5
Wa, Wb = [((torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s).requires_grad_() for _ in range(2)]
ba, bb, wc = [((torch.rand(H, 512, device=dev, generator=g) * 2 - 1) * s).requires_grad_() for _ in range(3)]
bc = ((torch.rand(H, device=dev, generator=g) * 2 - 1) * s).requires_grad_()
ds = torch.randn(T, H, device=dev, generator=g)
x = torch.randn(T, 512, device=dev, generator=g).requires_grad_()
W = (torch.randn(2048, 512, device=dev, generator=g) * 0.04).requires_grad_()
dy = torch.randn(T, 2048, device=dev, generator=g)
for it in range(a.iters):
    sc = MF.gate_scores(E, Wa, ba, Wb, bb, wc, bc, p_drop=0.25, seed=7 + it)
    sc.backward(ds)
    y = MF.linear(x, W)
    y.backward(dy)
torch.cuda.synchronize()
print("ok")
