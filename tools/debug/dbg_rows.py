import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from madeleine_amd import functional as MF
from tests.test_split_range_gpu import make_x
from tests._util import t
dev = torch.device("cuda:0")
for kind in ("zero_bags", "rows_over_30_binades", "uniform"):
    T, K, N = 1200, 512, 512
    x = make_x(kind, T, K); W = 0.05 * t((N, K), "rng:w"); lb = 0.3 * t((N,), "rng:lb")
    g, b = 1 + 0.2 * t((N,), "rng:g"), 0.3 * t((N,), "rng:b")
    dy = t((T, N), "rng:dy") * torch.logspace(0, -2, T).unsqueeze(1)
    lv = [v.double().requires_grad_() for v in (x, W, lb, g, b)]
    pre = lv[0] @ lv[1].t() + lv[2]
    pre.retain_grad()
    ref = F.gelu(F.layer_norm(pre, (N,), lv[3], lv[4], 1e-5)); ref.backward(dy.double())
    dl = [v.to(dev).requires_grad_() for v in (x, W, lb, g, b)]
    img, sc, out = MF.preattn_block(dl[0], None, dl[1], dl[2], dl[3], dl[4], 1e-5, 0.0, 0, None, True)
    out.backward(dy.to(dev))
    dx, rdx = dl[0].grad.double().cpu(), lv[0].grad
    top = rdx.abs().amax(1); err = (dx - rdx).abs().amax(1)
    rel = err / top.clamp_min(1e-300)
    worst = torch.argsort(rel, descending=True)[:6]
    print(kind, "dx_row worst rows", worst.tolist(), [f"{float(rel[i]):.2e}" for i in worst], "x rowmax", [f"{float(x[i].abs().max()):.2e}" for i in worst],
          "ref dx top", [f"{float(top[i]):.2e}" for i in worst], "dpre rowmax", [f"{float(pre.grad[i].abs().max()):.2e}" for i in worst])
    print("   dW rel", float((dl[1].grad.double().cpu() - lv[1].grad).norm() / lv[1].grad.norm()), "median row rel", float(rel.median()))
