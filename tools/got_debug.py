import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from madeleine_amd import functional as MF
from tests._util import t
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
for (k, n, d) in ((2, 2, 128), (3, 5, 128), (3, 9, 128), (7, 7, 128), (4, 16, 16), (8, 32, 128), (4, 70, 128)):
    v = t((k, n, d), f"dbg:v{k}{n}"); q = t((k, n, d), f"dbg:q{k}{n}") + 0.7 * v
    for name, w in (("wd", (1.0, 0.0)), ("gw", (0.0, 1.0))):
        v64, q64 = v.double().requires_grad_(), q.double().requires_grad_()
        c = R.threshold_relu(R.cross_cost(v64, q64))
        wd = (c * R.ipot(c, 0.5, 30)).sum(); gw = R.gw_distance(v64, q64).sum()
        (w[0] * wd + w[1] * gw).backward()
        v32, q32 = v.clone().requires_grad_(), q.clone().requires_grad_()
        c = R.threshold_relu(R.cross_cost(v32, q32))
        wd32 = (c * R.ipot(c, 0.5, 30)).sum(); gw32 = R.gw_distance(v32, q32).sum()
        (w[0] * wd32 + w[1] * gw32).backward()
        vd, qd = v.to(dev).requires_grad_(), q.to(dev).requires_grad_()
        o = MF.got(vd, qd)
        (w[0] * o[0] + w[1] * o[1]).backward()
        ref = float(wd if name == "wd" else gw); mine = float(o[0] if name == "wd" else o[1]); r32 = float(wd32 if name == "wd" else gw32)
        print(f"k={k} n={n} d={d} {name}: val hip {mine:.7f} ref64 {ref:.7f} ref32 {r32:.7f} | dV hip-vs-64 {rel(vd.grad, v64.grad):.2e} (ref32-vs-64 {rel(v32.grad, v64.grad):.2e}) dQ {rel(qd.grad, q64.grad):.2e} ({rel(q32.grad, q64.grad):.2e})")
