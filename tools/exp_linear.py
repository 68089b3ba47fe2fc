"""Times the fp32 Linear kernels against the library GEMM at the pre_attn shapes of config 2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
T = 262144
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, K, need_dx) in ((512, 512, False), (512, 512, True), (2048, 512, True)):
    x = torch.randn(T, K, device=dev).requires_grad_(need_dx)
    W = (torch.randn(N, K, device=dev) * 0.03).requires_grad_()
    dy = torch.randn(T, N, device=dev)
    fl = 2 * T * N * K
    for name, fn in (("hip", MF.linear), ("lib", torch.nn.functional.linear)):
        y = fn(x, W)
        f = timeit(lambda: fn(x, W))
        ins = (x, W) if need_dx else (W,)
        b = timeit(lambda: torch.autograd.grad(y, ins, dy, retain_graph=True))
        nb = 2 if need_dx else 1
        print(f"N={N} K={K} dx={need_dx} {name}: fwd {f:.3f} ms ({fl/f/1e9:.0f} TF)  bwd {b:.3f} ms ({nb*fl/b/1e9:.0f} TF)")
