"""Host synchronisations in one step of the reference-shaped path: model forward + trainer.calculate_losses (InfoNCE + GOT) + backward,
config-3 geometry (5 stains with absent stains), after warm-up."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import torch
from types import SimpleNamespace
from madeleine_amd import GOT, InfoNCE, MADELEINE, calculate_losses
dev = torch.device("cuda:0")
B, M, N, Dm, _, _ = BN.CONFIGS["c3"]
mods = BN.MODS5[:M]
torch.manual_seed(42)
model = MADELEINE(BN.make_cfg(M, Dm)).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
feats = torch.randn(B, M, N, Dm, device=dev)
rates = torch.tensor([1.0, 0.46, 0.73, 0.73, 0.73])
labels = (torch.rand(B, M, generator=torch.Generator().manual_seed(77)) < rates).float()
labels[:, 0] = 1
feats = feats * labels.to(dev)[:, :, None, None]
data = {"feats": feats, "modality_labels": labels}
crit = InfoNCE(temperature=0.001)
args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)


def step():
    opt.zero_grad(set_to_none=True)
    embs, toks = model(data, device=dev)
    loss, _ = calculate_losses(mods[1:], crit, GOT, None, embs, toks, labels[:, 1:], args)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("trainer-shaped c3 step: %.2f ms" % (1e3 * (time.perf_counter() - t0) / 5))
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
    torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one step:", len(w))
for x in w[:12]:
    print("  ", x.filename.split("/")[-1], x.lineno, str(x.message)[:90])
