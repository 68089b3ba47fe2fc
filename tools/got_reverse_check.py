#!/usr/bin/env python3
"""fp64 emulation of the reverse sweep implemented in csrc/got.hip, checked against autograd of the oracle.
Validates the hand derivation (IPOT backward, GW chain, thresholds, normalisation) independent of the GPU."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R

torch.manual_seed(0)
dt = torch.float64


def ipot_fwd(C, inv_beta, iters):
    n = C.shape[0]
    A = torch.exp(-C * inv_beta)
    sig = torch.full((n,), 1.0 / n, dtype=dt)
    T = torch.ones(n, n, dtype=dt)
    Th, dh, sh = [], [], [sig.clone()]
    for t in range(iters):
        Q = A * T
        de = 1.0 / (n * (Q @ sig))
        sig = 1.0 / (n * (Q.t() @ de))
        T = de[:, None] * Q * sig[None, :]
        Th.append(T.clone()); dh.append(de.clone()); sh.append(sig.clone())
    return Th, dh, sh


def ipot_bwd(C, inv_beta, iters, Th, dh, sh, gT):
    n = C.shape[0]
    A = torch.exp(-C * inv_beta)
    gA = torch.zeros_like(C)
    gsig = torch.zeros(n, dtype=dt)
    gT = gT.clone()
    for t in range(iters, 0, -1):
        Tp = Th[t - 2] if t >= 2 else torch.ones(n, n, dtype=dt)
        dl, sg, so = dh[t - 1], sh[t], sh[t - 1]
        Q = A * Tp
        gq = gT * Q
        gdel = (gq * sg[None, :]).sum(1)
        u = (gq * dl[:, None]).sum(0)
        ga = -n * sg * sg * (gsig + u)
        gdel = gdel + (Q * ga[None, :]).sum(1)
        gr = -n * dl * dl * gdel
        gsig = (Q * gr[:, None]).sum(0)
        gQ = gT * dl[:, None] * sg[None, :] + ga[None, :] * dl[:, None] + gr[:, None] * so[None, :]
        gA = gA + gQ * Tp
        gT = gQ * A
    return -inv_beta * gA * A


def got_manual(V, Q):
    k, n, d = V.shape
    rV, rQ = V.norm(dim=2), Q.norm(dim=2)
    Vh, Qh = V / (rV[..., None] + 1e-12), Q / (rQ[..., None] + 1e-12)
    C0 = 1 - Vh @ Qh.transpose(1, 2)
    Cs0 = 1 - Vh @ Vh.transpose(1, 2)
    Ct0 = 1 - Qh @ Qh.transpose(1, 2)
    ex = [C0.min(), C0.max(), Cs0.min(), Cs0.max(), Ct0.min(), Ct0.max()]
    thr = [ex[0] + 0.1 * (ex[1] - ex[0]), ex[2] + 0.1 * (ex[3] - ex[2]), ex[4] + 0.1 * (ex[5] - ex[4])]
    wd_tot, gw_tot = 0.0, 0.0
    G0 = torch.zeros_like(C0); Gs = torch.zeros_like(C0); Gt = torch.zeros_like(C0)
    gthr = [0.0, 0.0, 0.0]
    for b in range(k):
        C = torch.relu(C0[b] - thr[0])
        Th, dh, sh = ipot_fwd(C, 2.0, 30)
        wd_tot += (C * Th[-1]).sum()
        gC = Th[-1] + ipot_bwd(C, 2.0, 30, Th, dh, sh, C.clone())
        m = torch.where(C0[b] - thr[0] > 0, gC, torch.zeros_like(gC))
        G0[b] = m; gthr[0] -= m.sum()
        # GW
        Cs, Ct = torch.relu(Cs0[b] - thr[1]), torch.relu(Ct0[b] - thr[2])
        rs, rt = (Cs ** 2).sum(1) / n, (Ct ** 2).sum(1) / n
        gam = torch.full((n, n), 1.0 / (n * n), dtype=dt)
        hist = []
        for o in range(5):
            Cg = rs[:, None] + rt[None, :] - 2 * Cs @ gam @ Ct
            Th, dh, sh = ipot_fwd(Cg, 10.0, 20)
            hist.append((Cg, Th, dh, sh, gam))
            gam = Th[-1]
        Cgf = rs[:, None] + rt[None, :] - 2 * Cs @ gam @ Ct
        gw_tot += (Cgf * gam).sum()
        gCs = torch.zeros_like(Cs); gCt = torch.zeros_like(Ct)
        grs = torch.zeros(n, dtype=dt); grt = torch.zeros(n, dtype=dt)
        G = gam.clone()
        for o in range(5, -1, -1):
            g_in = hist[o - 1][1][-1] if o >= 1 else None
            grs += G.sum(1); grt += G.sum(0)
            if g_in is not None:
                P1 = G @ Ct
                gCs -= 2 * P1 @ g_in.t()
                gCt -= 2 * G.t() @ (Cs @ g_in)
                gTg = -2 * Cs @ P1
                Cg, Th, dh, sh, _ = hist[o - 1]
                G = ipot_bwd(Cg, 10.0, 20, Th, dh, sh, gTg)
            else:
                inv = 1.0 / (n * n)
                P1 = G @ Ct
                gCs -= 2 * (P1.sum(1) * inv)[:, None]
                cs = Cs.sum(1) * inv
                gCt += (-2 * (G * cs[:, None]).sum(0))[:, None]
        a = gCs + (2.0 / n) * Cs * grs[:, None]
        z = gCt + (2.0 / n) * Ct * grt[:, None]
        a = torch.where(Cs0[b] - thr[1] > 0, a, torch.zeros_like(a))
        z = torch.where(Ct0[b] - thr[2] > 0, z, torch.zeros_like(z))
        Gs[b] = a; Gt[b] = z; gthr[1] -= a.sum(); gthr[2] -= z.sum()
    # extrema routing (even split over ties)
    for m, (raw, Gm) in enumerate(((C0, G0), (Cs0, Gs), (Ct0, Gt))):
        for which, coef in ((0, 0.9), (1, 0.1)):
            mask = raw == ex[2 * m + which]
            Gm += mask * (coef * gthr[m] / mask.sum())
    gVh = -(G0 @ Qh) - (Gs + Gs.transpose(1, 2)) @ Vh
    gQh = -(G0.transpose(1, 2) @ Vh) - (Gt + Gt.transpose(1, 2)) @ Qh

    def nb(x, xh, r, g):
        s = 1.0 / (r + 1e-12)
        return s[..., None] * g - ((xh * g).sum(-1) / r)[..., None] * xh
    return wd_tot + gw_tot, nb(V, Vh, rV, gVh), nb(Q, Qh, rQ, gQh)


for (k, n) in ((2, 5), (3, 9), (4, 16)):
    V = torch.randn(k, n, 16, dtype=dt)
    Q = torch.randn(k, n, 16, dtype=dt) + 0.7 * V
    v, q = V.clone().requires_grad_(), Q.clone().requires_grad_()
    ref = R.got(v, q)
    ref.backward()
    val, dV, dQ = got_manual(V, Q)
    ev = float((dV - v.grad).norm() / v.grad.norm())
    eq = float((dQ - q.grad).norm() / q.grad.norm())
    print(f"k={k} n={n}: loss {float(ref):.8f} manual {float(val):.8f}  rel err dV {ev:.2e} dQ {eq:.2e}")
    assert abs(float(val) - float(ref)) < 1e-9 * abs(float(ref)) and ev < 1e-7 and eq < 1e-7
print("reverse sweep derivation OK")
