#!/usr/bin/env python3
"""Is the pooling forward's run-to-run bimodality (0.32 vs 0.35 ms, profiles/r05n_pool_fwd_occupancy_and_rows_in_flight.txt) a matter of where
the image of E (and the partial-sum workspace) lies?  One process, ONE big buffer, the same image copied to different byte offsets
inside it; pool_fwd timed at each; then the same with freshly allocated images."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF  # noqa: E402

dev = torch.device("cuda:0")
BM, N, H = 64, 4096, 4
g = torch.Generator(device=dev).manual_seed(0)
E2 = torch.randn(BM * N, H * 512, device=dev, generator=g)
scores = torch.randn(BM * N, H, device=dev, generator=g)
img0 = MF.split_image(E2)
rows, K = img0.rows, img0.K
nfl = rows * K


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


big = torch.empty(nfl + (64 << 20) // 4, device=dev, dtype=torch.float32)
print("base address of the big buffer: 0x%x" % big.data_ptr())
for off in (0, 256, 4096, 8192, 65536, 1 << 20, 2 << 20, (2 << 20) + 8192, 4 << 20, 16 << 20, 32 << 20, (32 << 20) + 4096):
    v = big[off // 4: off // 4 + nfl].view(rows, K)
    v.copy_(img0.data[:rows])
    im = MF.SplitImage(v, img0.scale, rows, K)
    ms = [timed(lambda: MF.pool_fwd_img_raw(im, scores, BM, N, None, N)) for _ in range(2)]
    print("offset %10d (addr %% 2MiB = %8d): %.4f %.4f ms" % (off, (big.data_ptr() + off) % (2 << 20), ms[0], ms[1]))
keep = []
for i in range(6):
    keep.append(torch.empty((i + 1) * 3_000_001, device=dev))   # perturb the allocator
    im = MF.split_image(E2)
    ms = timed(lambda: MF.pool_fwd_img_raw(im, scores, BM, N, None, N))
    print("fresh image %d at 0x%x: %.4f ms" % (i, im.data.data_ptr(), ms))
