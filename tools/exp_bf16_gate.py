"""Times the bf16 gate forward with / without the activation stores, and the backward pieces."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
T, H = 262144, 4
g = torch.Generator(device=dev).manual_seed(0)
E = torch.randn(T, H * 512, device=dev, generator=g).to(torch.bfloat16)
Wa = torch.randn(H, 512, 512, device=dev, generator=g) * 0.04
Wb = torch.randn(H, 512, 512, device=dev, generator=g) * 0.04
ba = torch.zeros(H, 512, device=dev); bb = torch.zeros(H, 512, device=dev)
wc = torch.randn(H, 512, device=dev, generator=g) * 0.05; bc = torch.zeros(H, device=dev)
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for save in (False, True):
    for p in (0.0, 0.25):
        ms = timeit(lambda: MF.gate_fwd_raw(E, Wa, ba, Wb, bb, wc, bc, p, 7, None, None, save))
        print(f"bf16 gate fwd save_act={save} p={p}: {ms:.3f} ms  ({2*T*H*512*1024/ms/1e9:.0f} TF)")
sc, a, b = MF.gate_fwd_raw(E, Wa, ba, Wb, bb, wc, bc, 0.25, 7, None, None, True)
ds = torch.randn(T, H, device=dev)
dE = torch.zeros_like(E)
for acc in (0, 1):
    ms = timeit(lambda: MF.gate_bwd_raw(E, Wa, Wb, wc, a, b, ds, dE, acc, 0.25, 7, None, None))
    print(f"bf16 gate bwd accumulate={acc}: {ms:.3f} ms")
E32 = E.float()
ms = timeit(lambda: MF.gate_fwd_raw(E32, Wa, ba, Wb, bb, wc, bc, 0.25, 7, None, None, True))
print(f"fp32 gate fwd: {ms:.3f} ms")
