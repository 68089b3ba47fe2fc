"""Host synchronisations in one data-parallel step (process group present: RCCL at world size 1 under torchrun): model forward +
distributed.calculate_losses_dp + backward + FlatGradSync + AdamW, config-2 and config-3 geometry."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import torch
from types import SimpleNamespace
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF
D.init_from_env()
dev = torch.device("cuda:0")
for cfg, use_got in (("c2", False), ("c3", True)):
    B, M, N, Dm, _, _ = BN.CONFIGS[cfg]
    mods = BN.MODS5[:M]
    torch.manual_seed(42)
    model = MADELEINE(BN.make_cfg(M, Dm)).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
    gs = D.FlatGradSync(model, use_got)
    feats = torch.randn(B, M, N, Dm, device=dev)
    labels = torch.ones(B, M)
    if M > 2:
        labels = (torch.rand(B, M, generator=torch.Generator().manual_seed(77)) < torch.tensor([1.0, 0.46, 0.73, 0.73, 0.73])).float()
        labels[:, 0] = 1
        feats = feats * labels.to(dev)[:, :, None, None]
    data = {"feats": feats, "modality_labels": labels}
    crit = InfoNCE(temperature=0.001)
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
    hg = D.host_group()

    def step():
        opt.zero_grad(set_to_none=True)
        pending = D.all_gather_labels_async(labels[:, 1:], hg)
        embs, toks = model(data, device=dev)
        loss, _ = D.calculate_losses_dp(mods[1:], crit, MF.HipGotImpl if use_got else None, embs, toks, labels[:, 1:], args,
                                        labels_global_withoutHE=pending.wait(), use_local_loss=use_got)
        loss.backward()
        gs.all_reduce_mean()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        step()
        torch.cuda.set_sync_debug_mode("default")
    print(cfg, "backend", torch.distributed.get_backend(), "synchronising calls in one step:", len(w))
    for x in w[:12]:
        print("  ", x.filename.split("/")[-1], x.lineno, str(x.message)[:90])
    del model, opt, feats, data
    torch.cuda.empty_cache()
