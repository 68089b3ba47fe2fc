#!/usr/bin/env python3
"""A/B of two builds of the GOT kernels through the raw C ABI: python tools/got_ab.py OLD.so [NEW.so]
Prints value / gradient agreement and fwd / bwd times at several (k, n)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import _native


def load(path):
    L = ctypes.CDLL(path)
    for name, (res, args) in _native.SIGNATURES.items():
        if name.startswith("mdl_got"):
            f = getattr(L, name); f.restype = res; f.argtypes = args
    return L


def run(L, v, q, reps=2):
    k, n, d = v.shape
    dev = v.device
    ws = torch.empty(L.mdl_got_ws_bytes(k, n, d), dtype=torch.uint8, device=dev)
    out = torch.empty(2, device=dev); dv = torch.empty_like(v); dq = torch.empty_like(q)
    go = torch.ones(2, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    best = (1e9, 1e9)
    for _ in range(reps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rc = L.mdl_got_fwd(v.data_ptr(), q.data_ptr(), out.data_ptr(), None, None, k, n, d, ws.data_ptr(), st); assert rc == 0, rc
        e[1].record()
        rc = L.mdl_got_bwd(v.data_ptr(), q.data_ptr(), go.data_ptr(), dv.data_ptr(), dq.data_ptr(), k, n, d, ws.data_ptr(), st); assert rc == 0, rc
        e[2].record(); torch.cuda.synchronize()
        best = (min(best[0], e[0].elapsed_time(e[1])), min(best[1], e[1].elapsed_time(e[2])))
    # exchange time-out flag of the split sweeps (got_impl.inc: ws[g_gen + 1]; the global region ends with g_gen[4])
    nf = (ws.numel() - 64) // 4
    err = float(ws[:nf * 4].view(torch.float32)[nf - 3]) if nf >= 4 else 0.0
    return out.clone(), dv.clone(), dq.clone(), best, err


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


if __name__ == "__main__":
    import json
    paths = sys.argv[1:] or [_native.lib_path()]
    libs = [(os.path.basename(p), load(p)) for p in paths]
    dev = torch.device("cuda:0")
    geoms = json.loads(os.environ.get("GOT_GEOMS", "[[4,129],[32,160],[32,192],[32,256]]"))
    for (k, n) in geoms:
        g = torch.Generator(device=dev).manual_seed(k * 1000 + n)
        v = torch.randn(k, n, 128, device=dev, generator=g)
        q = torch.randn(k, n, 128, device=dev, generator=g) + 0.7 * v
        ref = None
        for name, L in libs:
            o, dv, dq, t, err = run(L, v, q, reps=3)
            if ref is None:
                ref = (o, dv, dq)
            print(f"k={k:3d} n={n:3d} {name:<24} wd {float(o[0]):.6f} gw {float(o[1]):.6f}  dV rel {rel(dv, ref[1]):.2e} dQ rel {rel(dq, ref[2]):.2e}   "
                  f"{t[0]:7.3f} + {t[1]:7.3f} ms" + (f"  EXCHANGE TIME-OUT FLAG {err}" if err != 0.0 and name.startswith("libmadeleine") else ""), flush=True)
