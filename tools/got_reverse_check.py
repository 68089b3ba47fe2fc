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


def ipot_bwd_onepass(C, inv_beta, iters, Th, dh, sh, gT, dtype=None, q_from_plan=False):
    """The reverse sweep as csrc/got_impl.inc runs it since round 5 (ipot_backward_h): ONE pass over Q_t = A . T_{t-1} per iteration and
    no matrix-sized state besides the accumulator H.  With Y_t := gQ_t . Q_t (elementwise) the recurrences of ipot_bwd collapse:
        gq_t = gT_t . Q_t = Y_{t+1} / (delta_t,i sigma_t,j)      (A . Q_t = Q_{t+1} / (delta_t sigma_t), gT_t = gQ_{t+1} . A)
        gdel_i = rowsum(Y_{t+1})_i / delta_t,i ;  u_j = colsum(Y_{t+1})_j / sigma_t,j        -> only the row / column SUMS of Y travel
        W_t = Q_t . (delta_t ga_t^T + gr_t sigma_{t-1}^T) ;  Y_t = Y_{t+1} + W_t ;  Y_{iters+1} = gT_in . T_iters
        dL/dC = -(1/beta) sum_t Y_t = -(1/beta) (iters Y_{iters+1} + sum_t t W_t)
    q_from_plan: Q_t = T_t / (delta_t sigma_t) from the stored plan (one matrix read per iteration) instead of A . T_{t-1}."""
    dtp = dtype or dt
    n = C.shape[0]
    A = torch.exp(-C.to(dtp) * inv_beta)
    Y0 = gT.to(dtp) * Th[iters - 1].to(dtp)
    H = iters * Y0
    R, Cc = Y0.sum(1), Y0.sum(0)
    gsig = torch.zeros(n, dtype=dtp)
    for t in range(iters, 0, -1):
        dl, sg, so = dh[t - 1].to(dtp), sh[t].to(dtp), sh[t - 1].to(dtp)
        if q_from_plan:
            Q = Th[t - 1].to(dtp) * (1.0 / dl)[:, None] * (1.0 / sg)[None, :]
        else:
            Q = A * (Th[t - 2].to(dtp) if t >= 2 else torch.ones(n, n, dtype=dtp))
        ga = -n * sg * (sg * gsig + Cc)
        rs, rs2 = Q @ ga, Q @ so
        gr = -n * dl * dl * (R / dl + rs)
        W = Q * (dl[:, None] * ga[None, :] + gr[:, None] * so[None, :])
        H = H + t * W
        gsig = Q.t() @ gr
        R = R + dl * rs + gr * rs2
        Cc = Cc + W.sum(0)
    return -inv_beta * H


def check_onepass():
    global dt
    for n, inv_beta, iters in ((7, 2.0, 30), (33, 10.0, 20), (96, 10.0, 20)):
        C = torch.rand(n, n, dtype=dt) * 0.8
        gT = torch.randn(n, n, dtype=dt)
        Th, dh, sh = ipot_fwd(C, inv_beta, iters)
        ref = ipot_bwd(C, inv_beta, iters, Th, dh, sh, gT)
        Cr = C.clone().requires_grad_()
        A = torch.exp(-Cr * inv_beta)
        sig = torch.full((n,), 1.0 / n, dtype=dt)
        T = torch.ones(n, n, dtype=dt)
        for _ in range(iters):
            Q = A * T
            de = 1.0 / (n * (Q @ sig))
            sig = 1.0 / (n * (Q.t() @ de))
            T = de[:, None] * Q * sig[None, :]
        (auto,) = torch.autograd.grad((T * gT).sum(), Cr)
        for qp in (False, True):
            new = ipot_bwd_onepass(C, inv_beta, iters, Th, dh, sh, gT, q_from_plan=qp)
            e1 = float((new - auto).norm() / auto.norm())
            e0 = float((ref - auto).norm() / auto.norm())
            # the same two sweeps in fp32 on fp32-rounded histories (what the kernels hold), against the fp64 autograd result
            f32 = lambda x: x.float()  # noqa: E731
            Th32, dh32, sh32 = [f32(x) for x in Th], [f32(x) for x in dh], [f32(x) for x in sh]
            new32 = ipot_bwd_onepass(f32(C), inv_beta, iters, Th32, dh32, sh32, f32(gT), dtype=torch.float32, q_from_plan=qp)
            e32 = float((new32.double() - auto).norm() / auto.norm())
            print(f"one-pass sweep n={n} iters={iters} q_from_plan={qp}: fp64 vs autograd {e1:.1e} (two-pass {e0:.1e}); fp32 arithmetic {e32:.1e}")
            assert e1 < 1e-9
    keep = dt
    dt = torch.float32
    try:
        C = torch.rand(96, 96) * 0.8
        gT = torch.randn(96, 96)
        Th, dh, sh = ipot_fwd(C, 10.0, 20)
        old32 = ipot_bwd(C, 10.0, 20, Th, dh, sh, gT)
    finally:
        dt = keep
    Cd, gTd = C.double(), gT.double()
    Thd, dhd, shd = ipot_fwd(Cd, 10.0, 20)
    ref = ipot_bwd(Cd, 10.0, 20, Thd, dhd, shd, gTd)
    print(f"two-pass sweep in fp32 (round-4 kernels' arithmetic), n=96: {float((old32.double() - ref).norm() / ref.norm()):.1e}")


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


check_onepass()
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
