import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN  # sets GPU_MAX_HW_QUEUES default 8 unless the environment has it
import torch
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
extra = []
for _ in range(int(os.environ.get("DBG_EXTRA_STREAMS", "0"))):   # streams that exist before the GOT lanes do (a prefetcher's, RCCL's ...)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        torch.zeros(8, device=dev).add_(1)
    extra.append(st)
c3 = BN.secondary_c3_leg(dev, D, MF, InfoNCE, MADELEINE)
torch.cuda.empty_cache()
c4 = BN.secondary_c4_rank_leg(dev, D, MF, InfoNCE, MADELEINE)
print("extra streams", len(extra), "queues", os.environ.get("GPU_MAX_HW_QUEUES"), "blocking", bool(os.environ.get("MADELEINE_BLOCKING_H2D")), "c3 ms", c3["ms_per_step"], "c4 ms", c4["ms_per_step"],
      "got sum", c4["got_ms_per_step_sum_over_stains"], "ceiling", round(c3["ms_per_step"] / c4["ms_per_step"], 4))
