"""Where do the small (< 20 us) launches of a config-2 step come from?  One step under torch.profiler with Python stacks: every
aten op that issues a fill / copy / index / elementwise kernel, grouped by the innermost madeleine_amd / bench frame that called it."""
import collections
import os
import sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from tests._util import MODS5

dev = torch.device("cuda:0")
B, M, N, Dm = 32, 2, 4096, 512
cfg = SimpleNamespace(MODALITIES=MODS5[:M], wsi_encoder="abmil", patch_embedding_dim=Dm, wsi_encoder_hidden_dim=512, activation="softmax", n_heads=4)
torch.manual_seed(42)
model = MADELEINE(cfg, stain_encoding=False).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
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


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
small = collections.Counter()
dur = collections.Counter()
names = collections.defaultdict(collections.Counter)


def ancestors(ev):
    out = []
    p = ev.cpu_parent
    while p is not None:
        out.append(p.name)
        p = p.cpu_parent
    return out


for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    ks = [k for k in ev.kernels if k.duration < 20]
    if not ks:
        continue
    frame = next((f for f in ev.stack if ("madeleine_amd/" in f or "bench" in f or "tools/" in f) and "_native" not in f), None)
    if frame is None:   # autograd engine thread: name the backward node (or the outermost op) instead
        anc = ancestors(ev)
        node = next((a for a in anc if a.startswith("autograd::engine::evaluate_function")), anc[-1] if anc else "(top level)")
        frame = node.replace("autograd::engine::evaluate_function: ", "bwd node ")
    key = (ev.name, frame.split("/root/repo/")[-1] if "/root/repo/" in frame else frame[-90:])
    small[key] += len(ks)
    dur[key] += sum(k.duration for k in ks)
    for k in ks:
        names[key][k.name[:60]] += 1
print("launches < 20 us in ONE config-2 step: %d, %.1f us" % (sum(small.values()), sum(dur.values())))
for key, n in small.most_common(80):
    print("%3d  %7.1f us  %-28s %s   [%s]" % (n, dur[key], key[0], key[1], ", ".join("%s x%d" % kv for kv in names[key].most_common(4))))
