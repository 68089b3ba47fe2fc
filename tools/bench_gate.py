#!/usr/bin/env python3
"""Times the gate kernels (fwd, bwd) of one or more builds of libmadeleine_amd.so at config-2 geometry.
    python tools/bench_gate.py [--lib path.so ...] [--tokens 262144] [--iters 5]
Interleaves the libraries round-robin (within-process A/B) and prints per-kernel ms and TFLOP/s."""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--lib", action="append", default=[])
ap.add_argument("--tokens", type=int, default=262144)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--p", type=float, default=0.25)
a = ap.parse_args()
libs = a.lib or [_native.lib_path()]
dev = torch.device("cuda:0")
T, H = a.tokens, 4
g = torch.Generator(device=dev).manual_seed(0)
E = torch.randn(T, H * 512, device=dev, generator=g)
s = 1 / 512 ** 0.5
Wa = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
Wb = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
ba, bb, wc = [(torch.rand(H, 512, device=dev, generator=g) * 2 - 1) * s for _ in range(3)]
bc = torch.zeros(H, device=dev)
scores = torch.empty(T, H, device=dev); aa = torch.empty(T, H, 512, device=dev); ab = torch.empty_like(aa)
ds = torch.randn(T, H, device=dev, generator=g); dE = torch.empty_like(E)
dWa, dWb = torch.empty_like(Wa), torch.empty_like(Wb); dba, dbb, dwc = [torch.empty(H, 512, device=dev) for _ in range(3)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
handles = []
for path in libs:
    h = ctypes.CDLL(path)
    for name in ("mdl_abmil_gate_fwd", "mdl_abmil_gate_bwd", "mdl_abmil_gate_fwd_ws_bytes", "mdl_abmil_gate_bwd_ws_bytes"):
        fn = getattr(h, name); fn.restype, fn.argtypes = _native.SIGNATURES[name]
    handles.append(h)
wsf = torch.empty(handles[0].mdl_abmil_gate_fwd_ws_bytes(T, H), dtype=torch.uint8, device=dev)
wsb = torch.empty(handles[0].mdl_abmil_gate_bwd_ws_bytes(T, H), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
res = {p: {"fwd": [], "bwd": []} for p in libs}
for it in range(a.iters + 1):
    for path, h in zip(libs, handles):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rc = h.mdl_abmil_gate_fwd(P(E), E.stride(0), P(Wa), P(ba), P(Wb), P(bb), P(wc), P(bc), P(scores), P(aa), P(ab), T, H, a.p, 7, None, None, P(wsf), st)
        assert rc == 0, rc
        e[1].record()
        rc = h.mdl_abmil_gate_bwd(P(E), E.stride(0), P(Wa), P(Wb), P(wc), P(aa), P(ab), P(ds), P(dE), 0, P(dWa), P(dWb), P(dba), P(dbb), P(dwc), None, T, H, a.p, 7, None, None, P(wsb), st)
        assert rc == 0, rc
        e[2].record()
        torch.cuda.synchronize()
        if it > 0:
            res[path]["fwd"].append(e[0].elapsed_time(e[1])); res[path]["bwd"].append(e[1].elapsed_time(e[2]))
flop = T * H * 2 * 512 * 1024
for path in libs:
    f = sorted(res[path]["fwd"])[len(res[path]["fwd"]) // 2]; b = sorted(res[path]["bwd"])[len(res[path]["bwd"]) // 2]
    print(f"{os.path.basename(path):<40} fwd {f:7.3f} ms {flop/f/1e9:6.1f} TF | bwd {b:7.3f} ms {2*flop/b/1e9:6.1f} TF | total {f+b:7.3f} ms {3*flop/(f+b)/1e9:6.1f} TF   (scores sum {float(scores.sum()):.4f})")
