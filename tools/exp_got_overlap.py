"""GOT of one c4 rank (4 stains, k = cases of this rank, n = min(k_global, 256) tokens): wall time of got_multi (one HIP stream per stain)
against the same problems run one after the other."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import functional as MF, distributed as D

dev = torch.device("cuda:0")
ks, ns = [14, 23, 23, 24], [112, 180, 185, 188]
g = torch.Generator(device=dev).manual_seed(0)
probs = []
for k, n in zip(ks, ns):
    v = torch.randn(k, n, 128, device=dev, generator=g).requires_grad_()
    q = (torch.randn(k, n, 128, device=dev, generator=g) + 0.7 * v.detach()).requires_grad_()
    probs.append((v, q))


def timed(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def multi():
    ext = D.got_local_extrema(probs, MF.HipGotImpl)
    o = D.got_multi(probs, MF.HipGotImpl, None, extrema=ext)
    o.sum().backward()


def serial():
    for v, q in probs:
        o = MF.got(v, q)
        (o[0] + o[1]).backward()


def one(i):
    v, q = probs[i]
    o = MF.got(v, q)
    (o[0] + o[1]).backward()


print("got_multi (4 streams): %.2f ms" % timed(multi, 3 if os.environ.get("GOT_ONLY_MULTI") else 5))
if os.environ.get("GOT_ONLY_MULTI"):
    sys.exit(0)
print("serial               : %.2f ms" % timed(serial))
for i in range(4):
    print("  stain %d alone (k=%d n=%d): %.2f ms" % (i, ks[i], ns[i], timed(lambda: one(i))))
