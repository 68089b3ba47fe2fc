"""Achieved error of the HIP path on every GOT-bearing golden fixture (tests/golden/{got,calculate_losses,full_step}.npz),
quantity by quantity -- the numbers behind the tolerances written in tests/ (DESIGN.md section 4).  Runs on the GPU box:
    python tools/parity_report.py > gpurun_out/parity_report.json
Test infrastructure (imports oracle/ for the recipe inputs only)."""
import json
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests._util import MODS5, golden, rel_err, t  # noqa: E402
from tests.test_model_gpu import build  # noqa: E402


def relabs(a, b):
    return abs(float(a) - float(b)) / abs(float(b))


def main():
    from madeleine_amd import GOT, InfoNCE, calculate_losses
    from madeleine_amd import distributed as D
    from madeleine_amd import functional as MF
    dev = torch.device("cuda:0")
    rep = {}
    g = golden("got")
    for k in (2, 7, 32):
        trial = int(g[f"k{k}/trial"])
        v0, q0 = t((k, 40, 128), f"got:v{k}:{trial}"), t((k, 40, 128), f"got:q{k}:{trial}")
        q0 = q0 + 0.7 * v0
        vd, qd = v0.to(dev).requires_grad_(), q0.to(dev).requires_grad_()
        torch.manual_seed(100 + k)
        loss = GOT(vd, qd, subsample=256)
        loss.backward()
        e = {"loss": relabs(loss, g[f"k{k}/loss"]), "dv_norm": relabs(vd.grad.norm(), g[f"k{k}/dv_norm"]),
             "dq_norm": relabs(qd.grad.norm(), g[f"k{k}/dq_norm"])}
        if k <= 7:
            e["dv"] = rel_err(vd.grad[:, :k], g[f"k{k}/dv"])
            e["dq"] = rel_err(qd.grad[:, :k], g[f"k{k}/dq"])
        else:
            e["dv"] = rel_err(vd.grad[:4, :k, :16], g[f"k{k}/dv"])
            e["dq"] = rel_err(qd.grad[:4, :k, :16], g[f"k{k}/dq"])
        rep[f"got/k{k}"] = e

    g = golden("calculate_losses")
    B, M, N = 6, 5, 12
    stains = MODS5[1:]
    he_e, he_t = t((B, 1, 512), "cl:he_e"), t((B, N, 128), "cl:he_t")
    labels = torch.from_numpy(g["labels"])
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.7)
    for path in ("calculate_losses", "calculate_losses_dp"):
        wsi = {"HE": he_e.unsqueeze(3).repeat(1, 1, 1, M - 1).to(dev).requires_grad_()}
        tok = {"HE": he_t.unsqueeze(3).repeat(1, 1, 1, M - 1).to(dev).requires_grad_()}
        for s in stains:
            wsi[s] = (t((B, 1, 512), f"cl:e{s}") + 0.1 * he_e).to(dev).requires_grad_()
            tok[s] = (t((B, N, 128), f"cl:t{s}") + 0.6 * he_t).to(dev).requires_grad_()
        wx, tx = wsi, tok
        torch.manual_seed(5)
        if path == "calculate_losses":
            loss, _ = calculate_losses(stains, InfoNCE(temperature=0.001), GOT, None, wx, tx, labels[:, 1:], args)
        else:
            loss, _ = D.calculate_losses_dp(stains, InfoNCE(temperature=0.001), MF.HipGotImpl, wx, tx, labels[:, 1:], args)
        loss.backward()
        e = {"loss": relabs(loss, g["full/loss"])}
        for k in ["HE"] + stains:
            for kind, d in (("dwsi_norm", wsi), ("dtok_norm", tok)):
                ref = float(g[f"full/{kind}/{k}"])
                if ref > 0:
                    e[f"{kind}/{k}"] = abs(float(d[k].grad.norm()) - ref) / ref
        rep[f"{path}/full"] = e

    g = golden("full_step")
    B, M, N, Dm = (int(x) for x in g["shape"])
    mods = MODS5[:M]
    model = build(mods, Dm, "wfs", dev).eval()
    feats = t((B, M, N, Dm), "fs:feats")
    labels = torch.from_numpy(g["labels"])
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
    for path in ("calculate_losses", "calculate_losses_dp"):
        for use_got, prefix, key in ((True, "", "loss"), (False, "global/", "global/loss")):
            embs, toks = model({"feats": feats}, device=dev, train=True)
            torch.manual_seed(11)
            if path == "calculate_losses":
                loss, _ = calculate_losses(mods[1:], InfoNCE(temperature=0.001), GOT if use_got else None, None, embs, toks,
                                           labels[:, 1:], args)
            else:
                loss, _ = D.calculate_losses_dp(mods[1:], InfoNCE(temperature=0.001), MF.HipGotImpl if use_got else None, embs,
                                                toks, labels[:, 1:], args, use_local_loss=use_got)
            model.zero_grad()
            loss.backward()
            top = max(float(g[f"{prefix}gnorm/{k}"]) for k, _ in model.named_parameters())
            worst_n, worst_h = 0.0, 0.0
            for k, p in model.named_parameters():
                ref_n = float(g[f"{prefix}gnorm/{k}"])
                got = p.grad if p.grad is not None else torch.zeros_like(p)
                if ref_n > 1e-4 * top:
                    worst_n = max(worst_n, abs(float(got.norm()) - ref_n) / ref_n)
                head = torch.from_numpy(g[f"{prefix}ghead/{k}"])
                if float(head.norm()) > 1e-4 * top:
                    worst_h = max(worst_h, rel_err(got.flatten()[:16], head))
            rep[f"full_step/{path}/{'got' if use_got else 'global'}"] = {"loss": relabs(loss, g[key]), "worst_gnorm": worst_n,
                                                                         "worst_ghead": worst_h}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
