import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madeleine_amd import functional as MF
from tests.test_hip_kernels import _gate_weights
from tests._util import t
import math
dev = torch.device("cuda:0")
BM, N, H, peak = 3, 700, 4, float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
E = t((BM, N, H * 512), "rng:E") * 1.5
wa, ba, wb, bb, wc, bc = _gate_weights(H, "rng:gw"); wc = wc * peak
dp = t((BM, H * 512), "rng:dp")
src = [E, wa, ba, wb, bb, wc, bc]
lv = [v.double().requires_grad_() for v in src]
x = lv[0].view(BM, N, H, 512)
a = torch.tanh(torch.einsum("bnhe,hfe->bnhf", x, lv[1]) + lv[2]); b = torch.sigmoid(torch.einsum("bnhe,hfe->bnhf", x, lv[3]) + lv[4])
sc = ((a * b) * lv[5]).sum(-1) + lv[6]; sc.retain_grad()
w = torch.softmax(sc, dim=1)
ref = torch.einsum("bnh,bnhe->bhe", w, x).reshape(BM, H * 512); ref.backward(dp.double())
out = {}
for mode in ("split", "fp32"):
    MF.set_gemm_mode(mode)
    dl = [v.to(dev).requires_grad_() for v in src]
    pooled, scores = MF.attn_pool(*dl); pooled.backward(dp.to(dev))
    out[mode] = dl[0].grad.double().cpu().view(BM * N, H, 512)
r = lv[0].grad.view(BM * N, H, 512)
ds = sc.grad.view(BM * N, H)
for h in range(H):
    top = r[:, h].abs().amax(1); g = float(top.max())
    es, ef = (out["split"][:, h] - r[:, h]).abs().amax(1), (out["fp32"][:, h] - r[:, h]).abs().amax(1)
    idx = torch.argsort(es, descending=True)[:4]
    print("head", h, "global top %.2e  max|ds| %.2e" % (g, float(ds[:, h].abs().max())))
    for i in idx:
        print("   row %4d top/g 2^%.1f  |ds|/max 2^%.1f  err_split/g 2^%.1f  err_f32/g 2^%.1f  w %.2e" % (
            int(i), math.log2(float(top[i]) / g), math.log2(float(ds[i, h].abs() / ds[:, h].abs().max()) + 1e-300), math.log2(float(es[i]) / g + 1e-300),
            math.log2(float(ef[i]) / g + 1e-300), float(w.view(BM * N, H)[i, h])))
