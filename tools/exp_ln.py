"""Times the fused LayerNorm-GELU-Dropout kernels (fp32 / bf16 storage) and prints achieved HBM GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
T = 262144
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for W in (512, 2048):
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(T, W, device=dev).to(dt).requires_grad_()
        g = torch.ones(W, device=dev, requires_grad=True); b = torch.zeros(W, device=dev, requires_grad=True)
        P = float(os.environ.get("LN_P", "0.1")); y = MF.ln_gelu_drop(x, g, b, 1e-5, P, 5, None)
        dy = torch.randn_like(y)
        es = x.element_size()
        f = timeit(lambda: MF.ln_gelu_drop(x, g, b, 1e-5, P, 5, None))
        bw = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True))
        print(f"W={W} {str(dt):15s} fwd {f:.3f} ms ({2*T*W*es/f/1e6:.0f} GB/s)  bwd {bw:.3f} ms ({3*T*W*es/bw/1e6:.0f} GB/s)")
