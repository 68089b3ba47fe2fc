"""Per-step wall times of the config-4 rank leg (bench.secondary_c4_rank_leg's step), allocator statistics around them: looking for the
intermittent 2-6x slow timed passes seen in one of four processes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import torch
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
times = []
orig = BN.measure_leg


def measure(MF_, stepf, steps, warmup=2, prof_steps=3):
    for i in range(warmup):
        stepf()
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats(dev)
    t_all = time.perf_counter()
    for i in range(12):
        t0 = time.perf_counter()
        loss = stepf()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        times.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    st1 = torch.cuda.memory_stats(dev)
    print("12 steps, each synchronised: host enqueue ms / step ms:", " ".join("%.1f/%.1f" % t for t in times))
    print("device allocs", st1["num_device_alloc"] - st0["num_device_alloc"], "frees", st1["num_device_free"] - st0["num_device_free"],
          "retries", st1["num_alloc_retries"] - st0["num_alloc_retries"], "reserved GiB", st1["reserved_bytes.all.current"] / 2 ** 30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6):
        loss = stepf()
    torch.cuda.synchronize()
    print("6 steps back to back: %.1f ms/step" % (1e3 * (time.perf_counter() - t0) / 6))
    return orig(MF_, stepf, steps, warmup, prof_steps)


BN.measure_leg = measure
c4 = BN.secondary_c4_rank_leg(dev, D, MF, InfoNCE, MADELEINE)
print("c4 leg", c4["ms_per_step"])
