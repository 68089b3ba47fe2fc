"""Which small aten ops (fills, copies, index ops) does one bench step issue, and from where?  (torch profiler, python stacks)"""
import sys
from collections import Counter
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--no-extra-legs", "--no-bf16-leg"] + sys.argv[1:]
sys.path.insert(0, ".")
import bench
with profile(activities=[ProfilerActivity.CPU], record_shapes=False, with_stack=True) as prof:
    bench.main()
cnt = Counter()
for e in prof.events():
    if e.name in ("aten::zeros", "aten::zeros_like", "aten::fill_", "aten::zero_", "aten::copy_", "aten::index_select", "aten::cat", "aten::stack",
                  "aten::to", "aten::_to_copy", "aten::ones", "aten::full", "aten::arange", "aten::nonzero", "aten::item", "aten::_local_scalar_dense"):
        fr = [f for f in (e.stack or []) if "madeleine_amd" in f or "bench.py" in f]
        cnt[(e.name, fr[0] if fr else "?")] += 1
for (name, where), n in cnt.most_common(60):
    print(f"{n:5d}  {name:28s} {where}")
