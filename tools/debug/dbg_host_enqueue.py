"""Host-side enqueue time of the sections of the config-2 step (no synchronisation inside): a section that takes as long as the GPU
work queued before it is BLOCKING the host (e.g. a pageable H2D copy)."""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF
dev = torch.device("cuda:0")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
B, M, N, Dm, use_got, stain_enc = BN.CONFIGS[cfgname]
mods = BN.MODS5[:M]
torch.manual_seed(42)
model = MADELEINE(BN.make_cfg(M, Dm)).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
feats = torch.randn(B, M, N, Dm, device=dev)
labels = torch.ones(B, M)
if M > 2:
    rates = torch.tensor([1.0, 0.46, 0.73, 0.73, 0.73][:M])
    labels = (torch.rand(B, M, generator=torch.Generator().manual_seed(77)) < rates).float(); labels[:, 0] = 1
    feats = feats * labels.to(dev)[:, :, None, None]
data = {"feats": feats, "modality_labels": labels}
crit = InfoNCE(temperature=0.001)
largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
got_impl = MF.HipGotImpl if use_got else None
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc.setdefault(name, []).append(1e3 * (t - t0)); return t
for it in range(12):
    torch.cuda.synchronize()
    t = time.perf_counter(); t_start = t
    opt.zero_grad(set_to_none=True); t = tick("zero_grad", t)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=BF16):
        embs, toks = model(data, device=dev); t = tick("forward", t)
        loss, flag = D.calculate_losses_dp(mods[1:], crit, got_impl, embs, toks, labels[:, 1:], largs, use_local_loss=use_got); t = tick("loss", t)
    loss.backward(); t = tick("backward", t)
    opt.step(); t = tick("optimizer", t)
    acc.setdefault("host_total", []).append(1e3 * (t - t_start))
    torch.cuda.synchronize(); acc.setdefault("step_total", []).append(1e3 * (time.perf_counter() - t_start))
st = torch.cuda.memory_stats()
print("device mallocs", st.get("num_device_alloc"), "frees", st.get("num_device_free"), "reserved GiB", round(torch.cuda.memory_reserved() / 2**30, 2),
      "alloc retries", st.get("num_alloc_retries"))
for k, v in acc.items():
    v = sorted(v[2:]); print(f"{k:12s} median {v[len(v)//2]:8.3f} ms   min {v[0]:8.3f}")
