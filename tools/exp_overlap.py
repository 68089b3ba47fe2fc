"""Does an HBM-bound pass overlap an MFMA-bound contraction when both are queued on two HIP streams?  (round 3 experiment)
Times, at config-2 size: the 512->2048 Linear forward alone, the 2048-wide fused LN forward / backward alone, and both
queued concurrently on two streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T = 262144
x = torch.randn(T, 512, device=dev)
W = torch.randn(2048, 512, device=dev) * 0.04
y = torch.randn(T, 2048, device=dev)
g = torch.ones(2048, device=dev); b = torch.zeros(2048, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def gemm(n):
    for _ in range(n):
        MF.linear(x, W)

def ln(n):
    for _ in range(n):
        MF.ln_gelu_drop(y, g, b, 1e-5, 0.1, 7)

def timeit(fa, fb, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(s1):
            fa(n)
    if fb:
        with torch.cuda.stream(s2):
            fb(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n

with torch.no_grad():
    for _ in range(2):
        timeit(gemm, ln, 3)
    a = timeit(gemm, None); l = timeit(None, ln); both = timeit(gemm, ln)
    print(f"gemm alone {a:.3f} ms, LN alone {l:.3f} ms, both on two streams {both:.3f} ms (sum {a + l:.3f}, max {max(a, l):.3f})")
    # interleaved issue order (one of each at a time) to rule out queue-order effects
    def inter(n):
        for _ in range(n):
            with torch.cuda.stream(s1):
                MF.linear(x, W)
            with torch.cuda.stream(s2):
                MF.ln_gelu_drop(y, g, b, 1e-5, 0.1, 7)
    torch.cuda.synchronize(); t0 = time.perf_counter(); inter(10); torch.cuda.synchronize()
    print(f"interleaved issue: {(time.perf_counter() - t0) * 100:.3f} ms per pair")
    # 3 LN per GEMM (balanced durations)
    def ln3(n): ln(4 * n)
    l3 = timeit(None, ln3); both3 = timeit(gemm, ln3)
    print(f"gemm {a:.3f} + 4xLN {l3:.3f}: both {both3:.3f}")
