"""The c5 bf16 rank-emulation leg of bench.py alone, repeated (its time varied 66 ... 163 ms between default bench runs while the kernels'
sum stayed at 58 ms): is the variance in the leg (host / GOT exchange) or in what ran before it?  argv: repetitions [nosplit]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2 and sys.argv[2] == "nosplit":
    os.environ["MADELEINE_GOT_NOSPLIT"] = "1"
import torch
import bench
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    leg = bench.secondary_rank_leg(dev, D, MF, InfoNCE, MADELEINE, config="c5", steps=3, bf16=True)
    print("rep", rep, "ms_per_step", leg["ms_per_step"], "single_rank", leg.get("single_rank_ms_per_step"), "allocs", leg.get("device_allocs_in_timed_region"),
          "got", leg["got_ms_per_step_sum_over_stains"], flush=True)
    torch.cuda.empty_cache()
