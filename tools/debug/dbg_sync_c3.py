"""Which calls of a config-3 step synchronise the host with the device?  torch.cuda.set_sync_debug_mode('warn') around one step of
bench.secondary_c3_leg's loop (after warm-up)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import torch
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
state = {"n": 0}


def measure(MF_, stepf, steps, warmup=2, prof_steps=3):
    for _ in range(3):
        stepf()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        stepf()
        torch.cuda.set_sync_debug_mode("default")
    import traceback
    print("synchronising calls in one step:", len(w))
    for x in w[:12]:
        print("  ", x.filename.split("/")[-1], x.lineno, str(x.message)[:100])
    raise SystemExit(0)


BN.measure_leg = measure
BN.secondary_c3_leg(dev, D, MF, InfoNCE, MADELEINE)
