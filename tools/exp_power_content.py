"""Is the split NT product power-limited?  The same launch (262144 x 2048 x 512, the last pre_attn Linear of config 2) on random operands
and on operands whose bits barely toggle (constant matrices: hi plane constant, lo plane zero).  Same instructions, same memory traffic,
same schedule -- only the switching activity of the matrix cores and the data paths differs.  If the kernel were bound by its schedule
(LDS-DMA latency, barriers, issue) the two would take the same time; under a power cap the quiet operands run at a higher clock."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T, N, K = 262144, 2048, 512
torch.manual_seed(0)


def timed(a, b, iters=40):
    A, B = MF.split_image(a), MF.weight_image(b)
    for _ in range(5):
        MF.split_gemm_nt(A, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        MF.split_gemm_nt(A, B)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cases = {
    "random x random": (torch.randn(T, K, device=dev), 0.05 * torch.randn(N, K, device=dev)),
    "constant x constant (hi plane one value, lo plane zero)": (torch.full((T, K), 0.5, device=dev), torch.full((N, K), 0.25, device=dev)),
    "random x constant": (torch.randn(T, K, device=dev), torch.full((N, K), 0.25, device=dev)),
    "zero x random": (torch.zeros(T, K, device=dev), 0.05 * torch.randn(N, K, device=dev)),
}
flop = 2.0 * T * N * K * 3
for rep in range(2):
    for name, (a, b) in cases.items():
        ms = timed(a, b)
        print("%-58s %.3f ms  %.0f TFLOP/s raw (3 terms)" % (name, ms, flop / ms * 1e-9))
