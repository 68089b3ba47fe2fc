"""Does a captured graph of the config-2 step beat eager launches?  (measure instead of arguing.)
The step is captured ONCE with torch.cuda.CUDAGraph (hipGraph): the ctypes launches of libmadeleine_amd.so go to torch's current
stream, which is the capture stream inside the context.  Dropout seeds are host integers baked into the kernel arguments at capture, so
every replay draws the SAME masks -- fine for timing the launch tail, not a training mode (a real one needs device-side seeds).
Prints ms per step: eager, graph replay."""
import os
import sys
import time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from tests._util import MODS5

# the step's small host -> device copies (label-derived index tensors) go through the pinned-memory allocator, whose event queries are
# illegal during capture: memoise them by content (the labels of this experiment are constant)
from madeleine_amd import functional as MF
from madeleine_amd import trainer as TR
_cache, _h2d = {}, MF.h2d


def _cached_h2d(t, device, dtype=None):
    key = (t.dtype, tuple(t.shape), t.contiguous().numpy().tobytes(), str(dtype))
    if key not in _cache:
        _cache[key] = _h2d(t, device, dtype)
    return _cache[key]


for mod in (MF, D, TR):
    if hasattr(mod, "h2d"):
        mod.h2d = _cached_h2d

dev = torch.device("cuda:0")
B, M, N, Dm = 32, 2, 4096, 512
cfg = SimpleNamespace(MODALITIES=MODS5[:M], wsi_encoder="abmil", patch_embedding_dim=Dm, wsi_encoder_hidden_dim=512, activation="softmax", n_heads=4)
torch.manual_seed(42)
model = MADELEINE(cfg, stain_encoding=False).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
crit = InfoNCE(temperature=0.001)
largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
feats = torch.randn(B, M, N, Dm, device=dev)
labels = torch.ones(B, M)
data = {"feats": feats, "modality_labels": labels}


def step():
    opt.zero_grad(set_to_none=True)
    embs, toks = model(data, device=dev)
    loss, _ = D.calculate_losses_dp(MODS5[1:M], crit, None, embs, toks, labels[:, 1:], largs, labels_global_withoutHE=labels[:, 1:], use_local_loss=False)
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.synchronize()
eager = [timeit(step) for _ in range(3)]
print("eager          ms/step:", " ".join("%.3f" % v for v in eager), flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode=os.environ.get('CAPTURE_MODE', 'relaxed')):
        loss = step()
    torch.cuda.synchronize()
    rep = [timeit(g.replay) for _ in range(3)]
    print("graph replay   ms/step:", " ".join("%.3f" % v for v in rep), " loss", float(loss), flush=True)
    eager2 = [timeit(step) for _ in range(2)]
    print("eager again    ms/step:", " ".join("%.3f" % v for v in eager2), flush=True)
except Exception as e:      # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:600])
