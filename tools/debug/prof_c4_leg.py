"""One rank of config 4 (bench.secondary_c4_rank_leg) alone, for rocprofv3 --kernel-trace (set BENCH_NO_TIMER=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import torch
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
c4 = BN.secondary_c4_rank_leg(dev, D, MF, InfoNCE, MADELEINE)
print("c4 ms", c4["ms_per_step"], "got sum", c4.get("got_ms_per_step_sum_over_stains"))
