"""Which aten GEMM ops does one bench step still issue?  (torch profiler, shapes + python stack)"""
import sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--no-extra-legs"] + sys.argv[1:]
sys.path.insert(0, ".")
import bench
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.main()
seen = set()
for e in prof.events():
    if e.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::_scaled_mm", "aten::linear", "aten::matmul"):
        key = (e.name, str(e.input_shapes))
        if key in seen:
            continue
        seen.add(key)
        print(e.name, e.input_shapes)
        for fr in (e.stack or [])[:12]:
            print("    ", fr)
