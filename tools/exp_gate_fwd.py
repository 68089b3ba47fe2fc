import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from madeleine_amd import _native
import sys as _s
lib = _native.lib()
if len(_s.argv) > 1:
    lib = ctypes.CDLL(_s.argv[1])
    for n in ('mdl_abmil_gate_fwd','mdl_abmil_gate_fwd_ws_bytes'):
        f=getattr(lib,n); f.restype,f.argtypes=_native.SIGNATURES[n]
dev = torch.device("cuda:0"); T, H = 262144, 4
g = torch.Generator(device=dev).manual_seed(0)
E = torch.randn(T, H * 512, device=dev, generator=g); s = 1 / 512 ** 0.5
Wa = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s; Wb = (torch.rand(H, 512, 512, device=dev, generator=g) * 2 - 1) * s
ba, bb, wc = [(torch.rand(H, 512, device=dev, generator=g) * 2 - 1) * s for _ in range(3)]; bc = torch.zeros(H, device=dev)
scores = torch.empty(T, H, device=dev); aa = torch.empty(T, H, 512, device=dev); ab = torch.empty_like(aa)
ws = torch.empty(lib.mdl_abmil_gate_fwd_ws_bytes(T, H), dtype=torch.uint8, device=dev)
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr()); st = torch.cuda.current_stream().cuda_stream
for name, (A, B, p) in {"no act store p=0": (None, None, 0.0)}.items():
    ts = []
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.mdl_abmil_gate_fwd(P(E), E.stride(0), P(Wa), P(ba), P(Wb), P(bb), P(wc), P(bc), P(scores), P(A), P(B), T, H, p, 7, None, None, P(ws), st)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[1]; print(f"{name:<22} {t:7.3f} ms {T*H*2*512*1024/t/1e9:6.1f} TF")
