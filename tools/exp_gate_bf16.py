"""Lab: where the bf16 gate forward's time goes (config-2 geometry).  MADELEINE_GATE_LAB picks a stubbed epilogue variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import functional as MF

dev = torch.device("cuda:0")
T, H = int(os.environ.get("TOKENS", 262144)), 4
g = torch.Generator(device=dev).manual_seed(1)
E = torch.randn(T, H * 512, device=dev, generator=g).to(torch.bfloat16)
s = 1 / 512 ** 0.5
Wa, Wb = [((torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s) for _ in range(2)]
ba, bb, wc = [((torch.rand(H, 512, device=dev, generator=g) * 2 - 1) * s) for _ in range(3)]
bc = ((torch.rand(H, device=dev, generator=g) * 2 - 1) * s)


def run(p, save, n=20):
    if os.environ.get("MADELEINE_GATE_LAB") == "8":
        n = 1
    for _ in range(1 if n == 1 else 3):
        MF.gate_fwd_raw(E, Wa, ba, Wb, bb, wc, bc, p, 7, None, None, save)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        MF.gate_fwd_raw(E, Wa, ba, Wb, bb, wc, bc, p, 7, None, None, save)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


fl = 2.0 * T * H * 512 * 1024
for lab in os.environ.get("LABS", "0 1 2 3 4 5").split():
    os.environ["MADELEINE_GATE_LAB"] = lab
    ms = run(0.25, True)
    print("lab", lab, "p=.25 save: %.3f ms  %.3f PF" % (ms, fl / ms / 1e12))
os.environ["MADELEINE_GATE_LAB"] = "0"
for p, save in ((0.0, True), (0.25, False), (0.0, False)):
    ms = run(p, save)
    print("p=%.2f save=%d: %.3f ms  %.3f PF" % (p, save, ms, fl / ms / 1e12))
os.environ["MADELEINE_BF16_GATE128"] = "1"
print("128-tile p=.25 save: %.3f ms" % run(0.25, True))
for n in os.environ.get("SLEEPS", "0 1 2 3 4 6").split():
    os.environ["MADELEINE_GATE_SLEEP"] = n
    print("128-tile sleep", n, "p=.25 save: %.3f ms" % run(0.25, True))
