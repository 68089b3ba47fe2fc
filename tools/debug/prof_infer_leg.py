"""The inference leg alone (bench.secondary_inference_leg) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import torch
from madeleine_amd import MADELEINE
from madeleine_amd import functional as MF
r = BN.secondary_inference_leg(torch.device("cuda:0"), MF, MADELEINE)
print(r["value"], r["ms_per_bag"], r["kernel_ms"], r["bf16_autocast"])
