import sys, os, torch
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
from madeleine_amd import InfoNCE, MADELEINE
from madeleine_amd import distributed as D
from madeleine_amd import functional as MF
import madeleine_amd.model as MM
import torch.nn.functional as F
dev = torch.device("cuda:0")
MODS5 = ["HE", "HER2", "PGR", "KI67", "ER"]
def run(use_hip):
    torch.manual_seed(42)
    cfg = SimpleNamespace(MODALITIES=MODS5[:2], wsi_encoder="abmil", patch_embedding_dim=512, wsi_encoder_hidden_dim=512, activation="softmax", n_heads=4)
    model = MADELEINE(cfg).to(dev).train()
    if not use_hip:
        perm = model.wsi_embedders._perm
        model._project_slide = lambda p: F.linear(p, model.projector.weight[:, perm], model.projector.bias)
        model._project_tokens = lambda E: F.linear(E, model.token_projector.weight[:, perm], model.token_projector.bias)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    torch.manual_seed(1000)
    gen = torch.Generator(device=dev).manual_seed(1234)
    feats = torch.randn(32, 2, 4096, 512, device=dev, generator=gen)
    labels = torch.ones(32, 2)
    crit = InfoNCE(temperature=0.001)
    largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
    out = []
    for i in range(15):
        opt.zero_grad(set_to_none=True)
        embs, toks = model({"feats": feats, "modality_labels": labels}, device=dev)
        loss, _ = D.calculate_losses_dp(MODS5[1:2], crit, None, embs, toks, labels[:, 1:], largs, use_local_loss=False)
        loss.backward(); opt.step()
        out.append(float(loss.detach()))
    return out
a = run(True); b = run(False)
for i, (x, y) in enumerate(zip(a, b)): print(i, "%.6f %.6f" % (x, y))
