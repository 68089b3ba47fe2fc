"""Times the bf16 Linear kernels (mdl_linear_*_bf16) on the config-2 shapes and prints TFLOP/s per product."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
T = int(os.environ.get("LIN_T", 262144))
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for N, K in ((512, 512), (2048, 512), (128, 2048)):
    x = torch.randn(T, K, device=dev).to(torch.bfloat16).requires_grad_()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).requires_grad_()
    y = MF.linear(x, W)
    dy = torch.randn_like(y)
    gf = 2.0 * T * N * K / 1e9
    f = timeit(lambda: MF.linear(x, W))
    b = timeit(lambda: torch.autograd.grad(y, (x, W), dy, retain_graph=True))
    bw = timeit(lambda: torch.autograd.grad(y, (W,), dy, retain_graph=True))
    print(f"T={T} N={N} K={K}: fwd {f:.3f} ms ({gf/f:.0f} TF)  dX+dW {b:.3f} ms ({2*gf/b:.0f} TF)  dW only {bw:.3f} ms ({gf/bw:.0f} TF)")
