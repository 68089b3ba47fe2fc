"""Rate of the split NT product by shape at config-2 size (the five calls of a step are three shapes): is the short contraction (K = 512,
16 chunks per tile) or the wide output (N = 2048: 256 KiB of fp32 stores per 16 chunks) the slow one?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T = 262144
torch.manual_seed(0)


def timed(N, K, iters=30):
    a = torch.randn(T, K, device=dev)
    b = 0.05 * torch.randn(N, K, device=dev)
    A, B = MF.split_image(a), MF.weight_image(b)
    del a
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


for rep in range(2):
    for (N, K) in [(512, 512), (2048, 512), (512, 2048), (1024, 1024), (512, 1024), (256, 512), (2048, 2048)]:
        ms = timed(N, K)
        print("M %d N %4d K %4d  %.3f ms  %.0f TFLOP/s raw (3 terms)  C stores %.2f GB" % (T, N, K, ms, 2.0 * T * N * K * 3 / ms * 1e-9, T * N * 4e-9), flush=True)
