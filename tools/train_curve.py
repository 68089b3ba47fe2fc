"""Loss trajectory of a short pretrain run (fixed synthetic batch set, AdamW) in fp32 -- split GEMM mode (default) and exact-fp32
matrix-core mode -- and under bf16 autocast: python tools/train_curve.py [--steps 40].  All modes start from the same weights, see the
same batches and draw the same dropout masks."""
import argparse, os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madeleine_amd import GOT, InfoNCE, MADELEINE, calculate_losses
from madeleine_amd import functional as MF

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=40); a = ap.parse_args()
dev = torch.device("cuda:0")
mods = ["HE", "HER2", "PGR", "KI67", "ER"]
B, M, N, D = 16, 5, 512, 512
cfg = SimpleNamespace(MODALITIES=mods, wsi_encoder="abmil", patch_embedding_dim=D, wsi_encoder_hidden_dim=512,
                      activation="softmax", n_heads=4)
args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
g = torch.Generator().manual_seed(7)
base = torch.randn(4, B, 1, N, D, generator=g)                       # 4 batches; stains of a case share structure
batches = [(base[i] + 0.5 * torch.randn(B, M, N, D, generator=g)).to(dev) for i in range(4)]
labels = torch.ones(B, M)
curves = {}
for mode in ("float32", "float32-mfma", "float32-g2", "bfloat16"):
    MF.set_gemm_mode("fp32" if mode == "float32-mfma" else "split")
    MF.set_gradient_terms(2 if mode == "float32-g2" else 3)   # g2: the opt-in two-term backward products (forward unchanged)
    torch.manual_seed(42)
    model = MADELEINE(cfg).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    crit = InfoNCE(temperature=0.1)
    torch.manual_seed(1)
    out = []
    for step in range(a.steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=(mode == "bfloat16")):
            embs, toks = model({"feats": batches[step % 4]}, device=dev)
            loss, _ = calculate_losses(mods[1:], crit, GOT, None, embs, toks, labels[:, 1:], args)
        loss.backward()
        opt.step()
        out.append(float(loss.detach()))
    curves[mode] = out
MF.set_gemm_mode("split")
MF.set_gradient_terms(3)
for s in range(0, a.steps, max(1, a.steps // 10)):
    print(f"step {s:3d}  fp32(split) {curves['float32'][s]:10.5f}   fp32(mfma) {curves['float32-mfma'][s]:10.5f}   "
          f"split, 2-term bwd {curves['float32-g2'][s]:10.5f}   bf16 {curves['bfloat16'][s]:10.5f}")
print(f"last      fp32(split) {curves['float32'][-1]:10.5f}   fp32(mfma) {curves['float32-mfma'][-1]:10.5f}   "
      f"split, 2-term bwd {curves['float32-g2'][-1]:10.5f}   bf16 {curves['bfloat16'][-1]:10.5f}")
d2 = [abs(x - y) / max(abs(y), 1e-12) for x, y in zip(curves["float32-g2"], curves["float32"])]
print("two-term backward vs three terms: max relative loss difference over %d steps %.2e (first 10 steps: %.2e)" % (a.steps, max(d2), max(d2[:10])))
d = [abs(x - y) / max(abs(y), 1e-12) for x, y in zip(curves["float32"], curves["float32-mfma"])]
print("split vs exact-fp32 kernels: max relative loss difference over %d steps %.2e (first 10 steps: %.2e)" % (a.steps, max(d), max(d[:10])))
