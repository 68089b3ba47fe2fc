"""GPU parity tests proper: every HIP kernel, called through the C ABI (ctypes, via
madeleine_amd.functional), against the CPU oracle on the same seeded inputs.
Tolerance: 1e-3 relative fp32 (BASELINE.json north_star); most checks are far tighter."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests._util import max_rel, rel_err, t

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _hm(x, H=4):
    """reference channel order (e*H+c) -> head-major (c*512+e) on the last axis."""
    lead = x.shape[:-1]
    return x.reshape(*lead, 512, H).transpose(-1, -2).reshape(*lead, H * 512).contiguous()


def _gate_weights(H, key):
    s = 1.0 / np.sqrt(512.0)
    wa, wb = t((H, 512, 512), key + "wa", -s, s), t((H, 512, 512), key + "wb", -s, s)
    ba, bb = t((H, 512), key + "ba", -s, s), t((H, 512), key + "bb", -s, s)
    wc, bc = t((H, 512), key + "wc", -s, s), t((H,), key + "bc", -s, s)
    return wa, ba, wb, bb, wc, bc


def _oracle_scores(E_hm, w, keep_a=None, keep_b=None):
    """E_hm [T,H*512] head-major -> scores [T,H] with the oracle's per-head gate."""
    wa, ba, wb, bb, wc, bc = w
    H = wa.shape[0]
    T = E_hm.shape[0]
    x = E_hm.view(1, T, H, 512)
    out = []
    for c in range(H):
        ka = None if keep_a is None else keep_a[:, c].view(1, T, 512).float()
        kb = None if keep_b is None else keep_b[:, c].view(1, T, 512).float()
        out.append(R.gate_scores(x[:, :, c], wa[c], ba[c], wb[c], bb[c], wc[c:c + 1], bc[c:c + 1], ka, kb))
    return torch.cat(out, dim=-1).view(T, H)


# ---------------------------------------------------------------------------------------------- A3
@pytest.mark.parametrize("H", [4, 1])
@pytest.mark.parametrize("BM,N", [(3, 300), (2, 128), (1, 1), (2, 1000)])
def test_pool_fwd_bwd_dense(dev, H, BM, N):
    from madeleine_amd import functional as MF
    E = t((BM, N, H * 512), f"pool:E{BM}{N}{H}").requires_grad_()
    s = (t((BM, N, H), f"pool:s{BM}{N}{H}") * 6).requires_grad_()
    g = t((BM, H * 512), f"pool:g{BM}{N}{H}")
    w = torch.softmax(s, dim=1)                                      # [BM,N,H]
    ref = torch.einsum("bnh,bnhe->bhe", w, E.view(BM, N, H, 512)).reshape(BM, H * 512)
    ref.backward(g)
    Ed, sd = E.detach().to(dev).requires_grad_(), s.detach().to(dev).requires_grad_()
    out = MF.softmax_pool(Ed, sd)
    out.backward(g.to(dev))
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL
    assert rel_err(Ed.grad, E.grad) < 1e-5 and max_rel(Ed.grad, E.grad) < TOL
    assert float((sd.grad.cpu() - s.grad).abs().max()) <= 1e-4 * float(s.grad.abs().max()) + 1e-6


def test_pool_ragged_and_empty(dev):
    from madeleine_amd import functional as MF
    H = 4
    lens = [5, 0, 257, 128, 1]
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    T = int(cu[-1])
    E = t((T, H * 512), "rag:E").requires_grad_()
    s = (t((T, H), "rag:s") * 4).requires_grad_()
    g = t((len(lens), H * 512), "rag:g")
    refs = []
    for b, L in enumerate(lens):
        sl = slice(int(cu[b]), int(cu[b + 1]))
        if L == 0:
            refs.append(torch.zeros(H * 512))
            continue
        w = torch.softmax(s[sl], dim=0)
        refs.append(torch.einsum("nh,nhe->he", w, E[sl].view(L, H, 512)).reshape(-1))
    ref = torch.stack(refs)
    ref.backward(g)
    Ed, sd = E.detach().to(dev).requires_grad_(), s.detach().to(dev).requires_grad_()
    out = MF.softmax_pool(Ed, sd, cu.to(dev), max(lens))
    out.backward(g.to(dev))
    assert float(out[1].abs().max()) == 0.0
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL
    assert rel_err(Ed.grad, E.grad) < 1e-5 and max_rel(Ed.grad, E.grad) < TOL
    assert float((sd.grad.cpu() - s.grad).abs().max()) <= 1e-4 * float(s.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("bf16", [False, True])
def test_weighted_pool_no_softmax(dev, bf16):
    """mdl_abmil_wpool_*: the pooling of the relu / leaky_relu / sigmoid attention activations (abmil.py:56-61, Model.py:416-417 --
    weights used as they are, negative ones included), forward + both gradients, dense and ragged (with an empty bag),
    against the same einsum torch runs in the reference.  bf16 storage of E: one rounding of dE on the way out."""
    from madeleine_amd import functional as MF
    H = 4
    lens = [300, 0, 129, 1]
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    T = int(cu[-1])
    E = t((T, H * 512), "wp:E")
    if bf16:
        E = E.to(torch.bfloat16).float()
    E.requires_grad_()
    w = torch.nn.functional.leaky_relu(t((T, H), "wp:w") * 2).detach().requires_grad_()
    g = t((len(lens), H * 512), "wp:g")
    ref = torch.stack([torch.einsum("nh,nhe->he", w[int(cu[b]):int(cu[b + 1])],
                                    E[int(cu[b]):int(cu[b + 1])].view(L, H, 512)).reshape(-1) for b, L in enumerate(lens)])
    ref.backward(g)
    Ed = E.detach().to(dev).to(torch.bfloat16 if bf16 else torch.float32).requires_grad_()
    wd = w.detach().to(dev).requires_grad_()
    out = MF.weighted_pool(Ed, wd, cu.to(dev), max(lens))
    out.backward(g.to(dev))
    assert float(out[1].abs().max()) == 0.0
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL
    assert rel_err(Ed.grad.float(), E.grad) < (2.0 ** -8 if bf16 else 1e-5)
    assert rel_err(wd.grad, w.grad) < 1e-5
    # dense form [BM, N, H*512]
    E3 = t((3, 200, H * 512), "wp:E3").requires_grad_()
    w3 = torch.sigmoid(t((3, 200, H), "wp:w3")).detach().requires_grad_()
    ref3 = torch.einsum("bnh,bnhe->bhe", w3, E3.view(3, 200, H, 512)).reshape(3, -1)
    ref3.backward(g[:3])
    E3d, w3d = E3.detach().to(dev).requires_grad_(), w3.detach().to(dev).requires_grad_()
    out3 = MF.weighted_pool(E3d, w3d)
    out3.backward(g[:3].to(dev))
    assert rel_err(out3, ref3) < 1e-5 and rel_err(E3d.grad, E3.grad) < 1e-5 and rel_err(w3d.grad, w3.grad) < 1e-5


def test_pool_extreme_scores(dev):
    """softmax must be max-subtracted: scores of +-80 would overflow a naive exp."""
    from madeleine_amd import functional as MF
    E = t((2, 200, 2048), "ext:E")
    s = t((2, 200, 4), "ext:s") * 80
    ref = torch.einsum("bnh,bnhe->bhe", torch.softmax(s, dim=1), E.view(2, 200, 4, 512)).reshape(2, 2048)
    out = MF.softmax_pool(E.to(dev), s.to(dev))
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL


# ---------------------------------------------------------------------------------------------- A2
@pytest.mark.parametrize("H,T", [(4, 200), (1, 130), (4, 1), (2, 384)])
def test_gate_eval(dev, H, T):
    from madeleine_amd import functional as MF
    w = _gate_weights(H, f"gate{H}{T}")
    E = t((T, H * 512), f"gate:E{H}{T}")
    leaves = [x.clone().requires_grad_() for x in (E,) + w]
    ref = _oracle_scores(leaves[0], leaves[1:])
    g = t((T, H), f"gate:g{H}{T}")
    ref.backward(g)
    dl = [x.detach().to(dev).requires_grad_() for x in (E,) + w]
    out = MF.gate_scores(*dl)
    out.backward(g.to(dev))
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL
    names = ["E", "Wa", "ba", "Wb", "bb", "wc", "bc"]
    for n, a, b in zip(names, dl, leaves):
        assert rel_err(a.grad, b.grad) < 1e-4, n
        assert max_rel(a.grad, b.grad) < TOL, n


def test_gate_dropout_explicit_masks(dev):
    from madeleine_amd import functional as MF
    from oracle import recipe
    H, T = 4, 150
    w = _gate_weights(H, "gdo")
    E = t((T, H * 512), "gdo:E")
    ka = torch.from_numpy(recipe.bernoulli((T, H, 512), "gdo:ka", 0.75)).to(torch.uint8)
    kb = torch.from_numpy(recipe.bernoulli((T, H, 512), "gdo:kb", 0.75)).to(torch.uint8)
    leaves = [x.clone().requires_grad_() for x in (E,) + w]
    ref = _oracle_scores(leaves[0], leaves[1:], ka, kb)
    g = t((T, H), "gdo:g")
    ref.backward(g)
    dl = [x.detach().to(dev).requires_grad_() for x in (E,) + w]
    out = MF.gate_scores(*dl, p_drop=0.25, seed=0, keep_a=ka.to(dev), keep_b=kb.to(dev))
    out.backward(g.to(dev))
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL
    for n, a, b in zip(["E", "Wa", "ba", "Wb", "bb", "wc", "bc"], dl, leaves):
        assert rel_err(a.grad, b.grad) < 1e-4, n


def test_gate_dropout_rng_matches_exported_mask(dev):
    """The in-kernel counter RNG path == the explicit-mask path fed with the exported mask, bit for bit,
    forward and backward; keep-rate ~ 1-p; different seeds give different masks."""
    from madeleine_amd import _native
    from madeleine_amd import functional as MF
    H, T, p, seed = 4, 300, 0.25, 1234567890123
    lib = _native.lib()
    masks = []
    for which in (0, 1):
        m = torch.empty(T, H, 512, dtype=torch.uint8, device=dev)
        _native.check(lib.mdl_abmil_gate_dropout_mask(m.data_ptr(), T, H, which, p, seed,
                                                      torch.cuda.current_stream().cuda_stream), "mask")
        masks.append(m)
    rate = float(masks[0].float().mean())
    assert abs(rate - 0.75) < 0.01
    assert not torch.equal(masks[0], masks[1])
    w = [x.to(dev) for x in _gate_weights(H, "grng")]
    E = t((T, H * 512), "grng:E").to(dev)
    g = t((T, H), "grng:g").to(dev)
    res = []
    for kw in (dict(seed=seed), dict(seed=0, keep_a=masks[0], keep_b=masks[1])):
        leaves = [x.clone().requires_grad_() for x in [E] + w]
        out = MF.gate_scores(*leaves, p_drop=p, **kw)
        out.backward(g)
        res.append([out.detach()] + [x.grad for x in leaves])
    for a, b in zip(*res):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------- A2+A3 fused node
def test_attn_pool_fused_with_score_grad(dev):
    from madeleine_amd import functional as MF
    H, BM, N = 4, 3, 140
    w = _gate_weights(H, "ap")
    E = t((BM, N, H * 512), "ap:E")
    gp, gs = t((BM, H * 512), "ap:gp"), t((BM * N, H), "ap:gs") * 0.1
    leaves = [x.clone().requires_grad_() for x in (E,) + w]
    sc = _oracle_scores(leaves[0].view(BM * N, -1), leaves[1:])
    wts = torch.softmax(sc.view(BM, N, H), dim=1)
    pooled = torch.einsum("bnh,bnhe->bhe", wts, leaves[0].view(BM, N, H, 512)).reshape(BM, -1)
    ((pooled * gp).sum() + (sc * gs).sum()).backward()
    dl = [x.detach().to(dev).requires_grad_() for x in (E,) + w]
    po, so = MF.attn_pool(*dl)
    ((po * gp.to(dev)).sum() + (so * gs.to(dev)).sum()).backward()
    assert rel_err(po, pooled) < 1e-5 and rel_err(so, sc) < 1e-5
    for n, a, b in zip(["E", "Wa", "ba", "Wb", "bb", "wc", "bc"], dl, leaves):
        assert rel_err(a.grad, b.grad) < 1e-4, n


# ---------------------------------------------------------------------------------------------- L1
def _grad_ok(hip, ref32, ref64):
    """Within 1e-3 relative of the exact (fp64) gradient, or as accurate as the fp32 reference itself is.
    At T=0.001 a saturated softmax makes the CE gradient `softmax - onehot` a catastrophic cancellation: the
    fp32 reference is then only good to a few 10 % on gradients that are ~1e-6 of the usual scale, and
    agreement with it beyond its own error is meaningless."""
    e_hip = float((hip.detach().cpu().double() - ref64).norm())
    e_ref = float((ref32.double() - ref64).norm())
    assert e_hip <= max(TOL * float(ref64.norm()), 2.0 * e_ref) + 1e-30, (e_hip, e_ref, float(ref64.norm()))


@pytest.mark.parametrize("k", [2, 7, 33, 256])
@pytest.mark.parametrize("T", [0.001, 0.1])
@pytest.mark.parametrize("sym", [False, True])
def test_infonce_vs_oracle(dev, k, T, sym):
    from madeleine_amd import InfoNCE
    q0, p0 = t((k, 512), f"nce:q{k}"), t((k, 512), f"nce:p{k}")
    p0 = p0 + 0.1 * q0
    q, p = q0.clone().requires_grad_(), p0.clone().requires_grad_()
    ref = R.info_nce(q, p, T, sym)
    ref.backward()
    q64, p64 = q0.double().requires_grad_(), p0.double().requires_grad_()
    R.info_nce(q64, p64, T, sym).backward()
    qd, pd = q0.to(dev).requires_grad_(), p0.to(dev).requires_grad_()
    out = InfoNCE(temperature=T)(qd, pd, symmetric=sym)
    out.backward()
    assert abs(float(out) - float(ref)) <= TOL * abs(float(ref)) + 1e-5
    _grad_ok(qd.grad, q.grad, q64.grad)
    _grad_ok(pd.grad, p.grad, p64.grad)


@pytest.mark.parametrize("k,d", [(300, 64), (40, 1024)])
@pytest.mark.parametrize("sym", [False, True])
def test_infonce_staged_path(dev, k, d, sym):
    """More than 256 cases / an embedding wider than 512 (sizes the training loop never produces) against the fp64 oracle."""
    from madeleine_amd import InfoNCE
    T = 0.05
    q0, p0 = t((k, d), f"nces:q{k}"), t((k, d), f"nces:p{k}")
    p0 = p0 + 0.3 * q0
    q64, p64 = q0.double().requires_grad_(), p0.double().requires_grad_()
    ref = R.info_nce(q64, p64, T, sym)
    ref.backward()
    qd, pd = q0.to(dev).requires_grad_(), p0.to(dev).requires_grad_()
    out = InfoNCE(temperature=T)(qd, pd, symmetric=sym)
    out.backward()
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    assert rel_err(qd.grad, q64.grad) < 1e-4 and rel_err(pd.grad, p64.grad) < 1e-4


@pytest.mark.parametrize("sym", [False, True])
def test_infonce_fused_equals_staged_bitwise(dev, sym, monkeypatch):
    """The opt-in one-launch kernels (MADELEINE_INFONCE_FUSED=1; k <= 256, d <= 512) normalise on the fly with the staged path's
    products and summation orders: same bits for the losses and both gradients, ragged problem sizes and padding rows included."""
    from madeleine_amd import functional as MF
    S, Kmax, D = 3, 200, 512
    Q, P = t((S, Kmax, D), "ncef:q").to(dev), t((S, Kmax, D), "ncef:p").to(dev)
    P = P + 0.2 * Q
    cnt = torch.tensor([200, 77, 1], dtype=torch.int32, device=dev)
    Q[1, 77:] = float("nan")     # padding rows may hold anything
    res = []
    for fused in (False, True):
        if fused:
            monkeypatch.setenv("MADELEINE_INFONCE_FUSED", "1")
        q, p = Q.clone().requires_grad_(), P.clone().requires_grad_()
        loss = MF.info_nce_batched(q, p, cnt, 0.001, sym)
        (loss * torch.tensor([1.0, 2.0, 3.0], device=dev)).sum().backward()
        rows = MF.info_nce_rows(Q, P, cnt, 0.001, sym)
        res.append((loss.detach(), q.grad, p.grad, rows))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[0][1]).all() and torch.isfinite(res[0][2]).all()


def test_infonce_golden_and_batched(dev):
    from madeleine_amd import InfoNCE
    from tests._util import golden
    g = golden("infonce")
    crit = InfoNCE(temperature=0.001)
    ks = (2, 7, 33)
    Q = torch.zeros(3, 33, 512)
    P = torch.zeros(3, 33, 512)
    for s, k in enumerate(ks):
        q0, p0 = t((k, 512), f"nce:q{k}"), t((k, 512), f"nce:p{k}")
        Q[s, :k], P[s, :k] = q0, p0 + 0.1 * q0
    Qd, Pd = Q.to(dev).requires_grad_(), P.to(dev).requires_grad_()
    cnt = torch.tensor(ks, dtype=torch.int32, device=dev)
    losses = crit.batched(Qd, Pd, cnt, symmetric=True)
    losses.sum().backward()
    for s, k in enumerate(ks):
        tag = f"k{k}/T0.001/sym1"
        assert abs(float(losses[s]) - float(g[f"{tag}/loss"])) <= TOL * abs(float(g[f"{tag}/loss"])) + 1e-5
        sl = slice(None) if k <= 7 else slice(0, 32)
        if k >= 33:   # k = 2, 7 are saturated at T = 0.001: their reference gradients are rounding noise (see _grad_ok)
            assert rel_err(Qd.grad[s, :k, sl], g[f"{tag}/dq"]) < TOL
            assert rel_err(Pd.grad[s, :k, sl], g[f"{tag}/dp"]) < TOL
        else:
            q64, p64 = Q[s, :k].double().requires_grad_(), P[s, :k].double().requires_grad_()
            R.info_nce(q64, p64, 0.001, True).backward()
            _grad_ok(Qd.grad[s, :k], torch.from_numpy(g[f"{tag}/dq"]), q64.grad)
            _grad_ok(Pd.grad[s, :k], torch.from_numpy(g[f"{tag}/dp"]), p64.grad)
        assert float(Qd.grad[s, k:].abs().max() if k < 33 else 0.0) == 0.0


@pytest.mark.parametrize("sym", [False, True])
def test_infonce_reduction_none_sum_and_odd_width(dev, sym):
    """reduction 'none' / 'sum' (F.cross_entropy's other values, loss.py:58,124) and an embedding width that is not a multiple of
    32: per-sample losses and their gradients under a non-uniform upstream weight, against the oracle formula."""
    import torch.nn.functional as F
    from madeleine_amd import InfoNCE
    k, d, T = 19, 100, 0.07
    q0, p0 = t((k, d), "ncen:q"), t((k, d), "ncen:p")
    p0 = p0 + 0.2 * q0
    w = t((k,), "ncen:w").abs() + 0.1
    q, p = q0.double().requires_grad_(), p0.double().requires_grad_()
    logits = F.normalize(q, dim=-1) @ F.normalize(p, dim=-1).t() / T
    lab = torch.arange(k)
    ref = F.cross_entropy(logits, lab, reduction="none")
    if sym:
        ref = 0.5 * ref + 0.5 * F.cross_entropy(logits.t(), lab, reduction="none")
    (ref * w.double()).sum().backward()
    qd, pd = q0.to(dev).requires_grad_(), p0.to(dev).requires_grad_()
    out = InfoNCE(temperature=T, reduction="none")(qd, pd, symmetric=sym)
    assert tuple(out.shape) == (k,)
    (out * w.to(dev)).sum().backward()
    assert rel_err(out, ref) < 1e-5
    assert rel_err(qd.grad, q.grad) < 1e-4 and rel_err(pd.grad, p.grad) < 1e-4
    s = InfoNCE(temperature=T, reduction="sum")(q0.to(dev), p0.to(dev), symmetric=sym)
    assert abs(float(s) - float(ref.sum())) < 1e-5 * float(ref.sum())
    with pytest.raises(ValueError):
        InfoNCE(reduction="median")(q0.to(dev), p0.to(dev))


def test_infonce_errors(dev):
    from madeleine_amd import InfoNCE
    crit = InfoNCE()
    with pytest.raises(ValueError):
        crit(torch.zeros(2, 3, 4, device=dev), torch.zeros(2, 4, device=dev))
    with pytest.raises(ValueError):
        crit(torch.zeros(2, 32, device=dev), torch.zeros(3, 32, device=dev))
    with pytest.raises(ValueError):
        crit(torch.zeros(2, 32, device=dev), torch.zeros(2, 64, device=dev))
    with pytest.raises(RuntimeError):
        crit(torch.zeros(2, 32), torch.zeros(2, 32))  # CPU tensors: no fallback


# ---------------------------------------------------------------------------------------------- full sizes (C2)
def test_pool_full_size_properties(dev):
    """BASELINE config 2 geometry (64 bags x 4096 x 2048): size-independent properties + device-side reference."""
    from madeleine_amd import functional as MF
    BM, N, H = 64, 4096, 4
    gen = torch.Generator(device=dev).manual_seed(0)
    E = torch.randn(BM, N, H * 512, device=dev, generator=gen)
    s = torch.randn(BM, N, H, device=dev, generator=gen) * 3
    out = MF.softmax_pool(E, s)
    # (1) convexity: every pooled channel lies within [min_t, max_t] of its column
    assert bool((out <= E.amax(dim=1) + 1e-5).all()) and bool((out >= E.amin(dim=1) - 1e-5).all())
    # (2) shift invariance of the scores
    out2 = MF.softmax_pool(E, s + 7.5)
    assert rel_err(out2, out) < 1e-5
    # (3) a bag of identical tokens pools to that token whatever the scores
    Ec = E[:, :1].expand(-1, N, -1).contiguous()
    assert max_rel(MF.softmax_pool(Ec, s), Ec[:, 0]) < 1e-5
    # (4) linearity in E
    assert rel_err(MF.softmax_pool(2.5 * E, s), 2.5 * out) < 1e-6
    # (5) device-side fp32 restatement of the same op (chunked to bound memory)
    w = torch.softmax(s, dim=1)
    ref = torch.stack([torch.einsum("nh,nhe->he", w[b], E[b].view(N, H, 512)).reshape(-1) for b in range(BM)])
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL


# ---------------------------------------------------------------------------------------------- G0-G3
def _got_inputs(k, N, trial_key):
    from tests._util import golden
    g = golden("got")
    trial = int(g[f"k{k}/trial"])
    v0, q0 = t((k, N, 128), f"got:v{k}:{trial}"), t((k, N, 128), f"got:q{k}:{trial}")
    return g, v0, q0 + 0.7 * v0


@pytest.mark.parametrize("k", [2, 7, 32])
def test_got_vs_golden_and_oracle(dev, k):
    """GOT value + gradients against the vectors captured from the reference (well-conditioned instances, see
    oracle/gen_golden.py) -- including the randperm(batch) sub-sampling quirk: gradients land on tokens < k only."""
    from madeleine_amd import GOT
    N = 40
    g, v0, q0 = _got_inputs(k, N, "got")
    vd, qd = v0.to(dev).requires_grad_(), q0.to(dev).requires_grad_()
    torch.manual_seed(100 + k)                         # same randperm(k) draw as the golden run
    loss = GOT(vd, qd, subsample=256)
    loss.backward()
    # tolerances = 2x the error measured on MI355X per fixture (tools/parity_report.py, profiles/r02_parity_report.json):
    # loss <= 1.6e-6, gradient norms <= 1.4e-5, gradient tensors <= 6.2e-5 (k = 7) -- all far inside north_star's 1e-3
    ref = float(g[f"k{k}/loss"])
    assert abs(float(loss.detach()) - ref) < 4e-6 * abs(ref)
    assert abs(float(vd.grad.norm()) - float(g[f"k{k}/dv_norm"])) < 3e-5 * float(g[f"k{k}/dv_norm"])
    assert abs(float(qd.grad.norm()) - float(g[f"k{k}/dq_norm"])) < 3e-5 * float(g[f"k{k}/dq_norm"])
    assert float(vd.grad[:, k:].abs().max()) == 0.0
    if k <= 7:
        assert rel_err(vd.grad[:, :k], g[f"k{k}/dv"]) < 1.3e-4
        assert rel_err(qd.grad[:, :k], g[f"k{k}/dq"]) < 1.3e-4
    else:
        assert rel_err(vd.grad[:4, :k, :16], g[f"k{k}/dv"]) < 1.3e-4
        assert rel_err(qd.grad[:4, :k, :16], g[f"k{k}/dq"]) < 1.3e-4


def test_got_pieces_and_fp64_oracle(dev):
    """WD and GW sums separately against an fp64 evaluation of the oracle on a small well-conditioned problem."""
    from madeleine_amd import functional as MF
    from tests._util import golden
    g = golden("got")
    k, n = 3, 9
    trial = int(g["piece/trial"])
    v = t((k, n, 128), f"got:pv:{trial}")
    q = t((k, n, 128), f"got:pq:{trial}") + 0.5 * v
    out = MF.got(v.to(dev), q.to(dev))
    assert rel_err(out[0:1], g["piece/wd"].sum(keepdims=True).reshape(1)) < TOL
    assert rel_err(out[1:2], g["piece/gwd"].sum(keepdims=True).reshape(1)) < TOL
    v64, q64 = v.double().requires_grad_(), q.double().requires_grad_()
    R.got(v64, q64).backward()
    vd, qd = v.to(dev).requires_grad_(), q.to(dev).requires_grad_()
    o = MF.got(vd, qd)
    (o[0] + o[1]).backward()
    assert rel_err(vd.grad, v64.grad) < TOL and rel_err(qd.grad, q64.grad) < TOL


@pytest.mark.parametrize("k,n", [(2, 70), (3, 130), (2, 200), (2, 256), (2, 257), (2, 384), (1, 512)])
def test_got_large_n_vs_fp64_oracle(dev, k, n):
    """The IPOT / matrix-core paths of every n-class (n <= 64, <= 128, <= 256 register-resident plans; 256 < n <= 512 plans in
    the workspace, blocked products) against an fp64 evaluation of the oracle: both distances and the token gradients."""
    from madeleine_amd import functional as MF
    v = t((k, n, 128), f"got:big:v{n}:0")
    q = t((k, n, 128), f"got:big:q{n}:0") + 0.7 * v
    v64, q64 = v.double().requires_grad_(), q.double().requires_grad_()
    ref = R.got(v64, q64)
    ref.backward()
    vd, qd = v.to(dev).requires_grad_(), q.to(dev).requires_grad_()
    o = MF.got(vd, qd)
    (o[0] + o[1]).backward()
    assert abs(float(o.sum()) - float(ref)) < TOL * abs(float(ref))
    assert rel_err(vd.grad, v64.grad) < TOL and rel_err(qd.grad, q64.grad) < TOL


def test_got_api_without_subsample(dev):
    """GOT(v, q, subsample=None) (loss.py:278: every token of the bag enters the transport problem) with more tokens than the
    training loop's sub-sample: value and gradients against the fp64 oracle."""
    from madeleine_amd import GOT
    k, n = 2, 300
    v = t((k, n, 128), "got:api:v")
    q = t((k, n, 128), "got:api:q") + 0.7 * v
    v64, q64 = v.double().requires_grad_(), q.double().requires_grad_()
    ref = R.got(v64, q64, subsample=None)
    ref.backward()
    vd, qd = v.to(dev).requires_grad_(), q.to(dev).requires_grad_()
    loss = GOT(vd, qd, subsample=None)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref)) < TOL * abs(float(ref))
    assert rel_err(vd.grad, v64.grad) < TOL and rel_err(qd.grad, q64.grad) < TOL


def test_got_external_thresholds_and_limits(dev):
    """minmax_in = the batch's own extrema reproduces the local result (value and gradient); two half-batches with the
    global extrema sum to the full batch (the data-parallel decomposition); n > 512 is refused loudly."""
    from madeleine_amd import functional as MF
    k, n = 6, 12
    v = t((k, n, 128), "got:ext:v").to(dev)
    q = (t((k, n, 128), "got:ext:q") + 0.6 * t((k, n, 128), "got:ext:v")).to(dev)
    v1, q1 = v.clone().requires_grad_(), q.clone().requires_grad_()
    o1, mm = MF.got(v1, q1, return_extrema=True)
    (o1[0] + o1[1]).backward()
    parts, grads = [], []
    acc = {"d": torch.zeros(6, device=dev)}
    # pass 1: collect d_minmax of both halves (what the all-reduce would sum); pass 2: finish with the total
    halves = [(v[:3].clone().requires_grad_(), q[:3].clone().requires_grad_()),
              (v[3:].clone().requires_grad_(), q[3:].clone().requires_grad_())]
    dmm_parts = []
    for vh, qh in halves:
        o = MF.got(vh, qh, minmax_in=mm, reduce_dminmax=lambda x: (dmm_parts.append(x.clone()) or x))
        (o[0] + o[1]).backward()
        parts.append(o.detach())
    total = dmm_parts[0] + dmm_parts[1]
    for vh, qh in halves:
        vh.grad = None
        qh.grad = None
        o = MF.got(vh, qh, minmax_in=mm, reduce_dminmax=lambda x: total)
        (o[0] + o[1]).backward()
        grads.append((vh.grad, qh.grad))
    assert rel_err(parts[0] + parts[1], o1.detach()) < 1e-5
    assert rel_err(torch.cat([grads[0][0], grads[1][0]]), v1.grad) < 1e-4
    assert rel_err(torch.cat([grads[0][1], grads[1][1]]), q1.grad) < 1e-4
    with pytest.raises(NotImplementedError):
        MF.got(torch.zeros(1, 600, 128, device=dev), torch.zeros(1, 600, 128, device=dev))


# ---------------------------------------------------------------------------------------------- N1 fused LN-GELU-Dropout
@pytest.mark.parametrize("W,rows", [(512, 300), (2048, 77), (512, 1), (1024, 51), (4096, 9), (256, 13)])
@pytest.mark.parametrize("mode", ["eval", "mask", "mask+bias"])
def test_ln_gelu_drop_vs_torch(dev, W, rows, mode):
    """mask+bias: the preceding Linear's bias is added inside the kernel and its gradient (column sums of dx) comes out of
    the same backward pass."""
    import torch.nn.functional as F
    from madeleine_amd import functional as MF
    from oracle import recipe
    x = (t((rows, W), f"ln:x{W}{rows}") * 3 + 0.5).requires_grad_()
    g = (1 + 0.2 * t((W,), f"ln:g{W}")).requires_grad_()
    b = (0.3 * t((W,), f"ln:b{W}")).requires_grad_()
    lb = (0.7 * t((W,), f"ln:lb{W}")).requires_grad_() if mode == "mask+bias" else None
    dy = t((rows, W), f"ln:dy{W}{rows}")
    keep = torch.from_numpy(recipe.bernoulli((rows, W), f"ln:k{W}{rows}", 0.9)) if mode != "eval" else None
    ref = F.gelu(F.layer_norm(x if lb is None else x + lb, (W,), g, b, 1e-5))
    if keep is not None:
        ref = ref * keep / 0.9
    ref.backward(dy)
    xd, gd, bd = (v.detach().to(dev).requires_grad_() for v in (x, g, b))
    lbd = None if lb is None else lb.detach().to(dev).requires_grad_()
    out = MF.ln_gelu_drop(xd, gd, bd, 1e-5, 0.1 if keep is not None else 0.0, 0,
                          None if keep is None else keep.to(torch.uint8).to(dev), lbd)
    out.backward(dy.to(dev))
    assert rel_err(out, ref) < 1e-5 and max_rel(out, ref) < TOL
    assert rel_err(xd.grad, x.grad) < 1e-4
    assert rel_err(gd.grad, g.grad) < 1e-4 and rel_err(bd.grad, b.grad) < 1e-4
    if lb is not None:
        assert rel_err(lbd.grad, lb.grad) < 1e-4


@pytest.mark.parametrize("W,dtype", [(512, torch.float32), (512, torch.bfloat16), (256, torch.float32), (1024, torch.bfloat16)])
def test_ln_gelu_drop_groups_vs_torch(dev, W, dtype):
    """Round 6: the grouped LayerNorm-GELU-Dropout pass of the bf16 / exact-fp32 engines -- one bias row per GROUP of rows (the stain
    encoding folded out of the first Linear: Model.py:125-132, :351; [x | e_g] W^T = x Wx^T + e_g We^T) -- against torch in fp64 on the same
    (storage-rounded) input: output, dx, dgamma, dbeta and the per-group bias gradients; ragged groups incl. an empty one and one of a single
    row; injected keep mask and the in-kernel RNG (same seed -> same bits, group launch or not)."""
    import torch.nn.functional as F
    from madeleine_amd import functional as MF
    from oracle import recipe
    lens = [37, 0, 1, 300, 5, 129]
    G, rows = len(lens), sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int64)
    x = (t((rows, W), f"lng:x{W}") * 3 + 0.5).to(dtype)
    g = (1 + 0.2 * t((W,), f"lng:g{W}")).requires_grad_()
    b = (0.3 * t((W,), f"lng:b{W}")).requires_grad_()
    gb = (0.7 * t((G, W), f"lng:gb{W}")).requires_grad_()
    dy = t((rows, W), f"lng:dy{W}").to(dtype)
    keep = torch.from_numpy(recipe.bernoulli((rows, W), f"lng:k{W}", 0.9))
    x64 = x.double().requires_grad_()
    row_group = torch.repeat_interleave(torch.arange(G), torch.tensor(lens))
    ref = F.gelu(F.layer_norm(x64 + gb.double()[row_group], (W,), g.double(), b.double(), 1e-5)) * keep / 0.9
    ref.backward(dy.double())
    xd = x.to(dev).requires_grad_()
    gd, bd, gbd = (v.detach().to(dev).requires_grad_() for v in (g, b, gb))
    out = MF.ln_gelu_drop_groups(xd, gd, bd, 1e-5, 0.1, 0, keep.to(torch.uint8).to(dev), gbd, cu.to(dev))
    out.backward(dy.to(dev))
    lo = dtype == torch.bfloat16
    tol_o, tol_g = (6e-3, 1e-2) if lo else (1e-5, 1e-4)      # bf16: the storage grid of y / dx (2^-9 relative per element)
    assert rel_err(out.float(), ref.float()) < tol_o
    assert rel_err(xd.grad.float(), x64.grad.float()) < tol_g
    assert rel_err(gd.grad, g.grad) < tol_g and rel_err(bd.grad, b.grad) < tol_g
    assert rel_err(gbd.grad, gb.grad) < tol_g
    assert float(gbd.grad[1].abs().max()) == 0.0                     # the empty group
    # the same numbers as the un-grouped kernel given the gathered bias rows added beforehand (fp32 storage: exact same arithmetic)
    if not lo:
        x2 = (x.to(dev) + gb.detach().to(dev)[row_group.to(dev)]).requires_grad_()
        o2 = MF.ln_gelu_drop(x2, gd.detach(), bd.detach(), 1e-5, 0.1, 0, keep.to(torch.uint8).to(dev))
        assert rel_err(out, o2) < 1e-6
    # in-kernel RNG: row-indexed counters, so the group launch draws the mask of the plain launch
    o_r = MF.ln_gelu_drop_groups(xd.detach(), gd.detach(), bd.detach(), 1e-5, 0.1, 1234, None, torch.zeros(G, W, device=dev), cu.to(dev))
    o_p = MF.ln_gelu_drop(xd.detach(), gd.detach(), bd.detach(), 1e-5, 0.1, 1234)
    assert torch.equal(o_r, o_p)


def test_ln_gelu_drop_rng_is_consistent(dev):
    """in-kernel RNG: keep rate ~0.9, same seed -> same output, backward zeros exactly where forward dropped."""
    from madeleine_amd import functional as MF
    W, rows = 2048, 200
    x = t((rows, W), "lnr:x").to(dev).requires_grad_()
    g, b = torch.ones(W, device=dev), torch.zeros(W, device=dev)
    y1 = MF.ln_gelu_drop(x, g, b, 1e-5, 0.1, 99)
    y2 = MF.ln_gelu_drop(x, g, b, 1e-5, 0.1, 99)
    y3 = MF.ln_gelu_drop(x, g, b, 1e-5, 0.1, 100)
    y0 = MF.ln_gelu_drop(x, g, b, 1e-5, 0.0, 0)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    dropped = (y1 == 0) & (y0 != 0)
    assert abs(float(dropped.float().mean()) - 0.1) < 0.01
    assert rel_err(y1[~dropped], (y0 / 0.9)[~dropped]) < 1e-6
    y1.backward(torch.ones_like(y1))
    # an element dropped in forward contributes no gradient path: check via a one-hot upstream gradient
    x.grad = None
    yy = MF.ln_gelu_drop(x, g, b, 1e-5, 0.1, 99)
    r, c = [int(v[0]) for v in dropped.nonzero(as_tuple=True)]
    sel = torch.zeros_like(yy)
    sel[r, c] = 1.0
    yy.backward(sel)
    assert float(x.grad.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------- N1 Linear (fp32 MFMA)
@pytest.mark.parametrize("T,N,K,need_dx", [(300, 512, 512, True), (1000, 2048, 512, True), (77, 512, 544, False),
                                           (130, 512, 800, False), (5, 256, 256, True),
                                           # dX with an output width K that is not a multiple of the 256-column tile (config 5's
                                           # 768 + 32 channels; Model.py:132 makes that input require grad): ragged-column tile
                                           (1000, 512, 800, True), (300, 512, 544, True), (513, 256, 32, True), (2500, 512, 1056, True)])
def test_linear_vs_torch(dev, T, N, K, need_dx):
    """Bias-free Linear fwd / dX / dW on the fp32 matrix cores against torch in fp64 (ragged T: tile and split tails; K not a
    multiple of the 128-row dW tile; transposes detected by the asymmetric shapes)."""
    from madeleine_amd import functional as MF
    x = t((T, K), f"lin:x{T}{K}")
    W = 0.05 * t((N, K), f"lin:w{N}{K}")
    dy = t((T, N), f"lin:dy{T}{N}")
    x64, W64 = x.double().requires_grad_(), W.double().requires_grad_()
    (x64 @ W64.t()).backward(dy.double())
    xd = x.to(dev).requires_grad_(need_dx)
    Wd = W.to(dev).requires_grad_()
    assert MF.linear_supported(xd, Wd)
    y = MF.linear(xd, Wd)
    y.backward(dy.to(dev))
    assert rel_err(y, x64.detach() @ W64.detach().t()) < 1e-6
    assert rel_err(Wd.grad, W64.grad) < 1e-6
    if need_dx:
        assert rel_err(xd.grad, x64.grad) < 1e-6


@pytest.mark.parametrize("T,N,K,need_dx", [(1000, 128, 2048, True), (300, 384, 256, True), (64, 512, 2048, True), (192, 512, 2048, True),
                                           (7, 12, 100, True), (700, 128, 512, False)])
def test_linear_bias_tall_and_small_paths(dev, T, N, K, need_dx):
    """token_projector-like (N = 128 (mod 256): tall 256 x 128 tile, role-swapped dW), projector-like (T <= 256 rows: fp32 FMA
    kernel) and the bias / dbias path, against torch in fp64."""
    from madeleine_amd import functional as MF
    x = t((T, K), f"linb:x{T}{K}")
    W = 0.05 * t((N, K), f"linb:w{N}{K}")
    b = 0.3 * t((N,), f"linb:b{N}")
    dy = t((T, N), f"linb:dy{T}{N}")
    x64, W64, b64 = x.double().requires_grad_(), W.double().requires_grad_(), b.double().requires_grad_()
    (x64 @ W64.t() + b64).backward(dy.double())
    xd = x.to(dev).requires_grad_(need_dx)
    Wd, bd = W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    assert MF.linear_supported(xd, Wd)
    y = MF.linear(xd, Wd, bd)
    y.backward(dy.to(dev))
    assert rel_err(y, (x64 @ W64.t() + b64).detach()) < 1e-6
    assert rel_err(Wd.grad, W64.grad) < 1e-6 and rel_err(bd.grad, b64.grad) < 1e-6
    if need_dx:
        assert rel_err(xd.grad, x64.grad) < 1e-6


def test_linear_unsupported_geometry_raises(dev):
    """No library-GEMM fallback in the product path: a geometry outside the HIP kernels raises."""
    from madeleine_amd import functional as MF
    x, W = t((300, 100), "lin:ux").to(dev), t((130, 100), "lin:uw").to(dev)
    assert not MF.linear_supported(x, W)
    with pytest.raises(NotImplementedError):
        MF.linear(x, W)
    with pytest.raises(NotImplementedError):
        MF.linear(t((300, 100), "lin:ux"), t((256, 100), "lin:uw2"))      # CPU tensors: no eager path either


def test_linear_few_rows_bf16(dev):
    """A handful of bf16 rows (one short bag under autocast) run the exact fp32 small-M kernel and store bf16."""
    from madeleine_amd import functional as MF
    x = t((40, 512), "lin:fx").to(dev).to(torch.bfloat16).requires_grad_()
    W = (0.05 * t((512, 512), "lin:fw")).to(dev).requires_grad_()
    y = MF.linear(x, W)
    assert y.dtype == torch.bfloat16
    y.float().sum().backward()
    ref = x.detach().float() @ W.detach().t()
    assert rel_err(y.float(), ref) < 2.0 ** -8
    assert rel_err(W.grad, torch.ones(40, 512, device=dev).t() @ x.detach().float()) < 1e-5


def test_linear_full_size_vs_library(dev):
    """Config-2 size (T = 262,144 tokens, 512 -> 2048): forward, dX and dW against the library GEMM on the device."""
    from madeleine_amd import functional as MF
    T, N, K = 262144, 2048, 512
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(T, K, device=dev, generator=g).requires_grad_()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.04).requires_grad_()
    dy = torch.randn(T, N, device=dev, generator=g)
    y = MF.linear(x, W)
    gx, gW = torch.autograd.grad(y, (x, W), dy)
    yr = torch.nn.functional.linear(x, W)
    rx, rW = torch.autograd.grad(yr, (x, W), dy)
    assert rel_err(y[::997], yr[::997]) < 1e-5 and rel_err(gx[::997], rx[::997]) < 1e-5
    assert rel_err(gW, rW) < 1e-4      # 262,144-term fp32 sums in two different orders


def test_linear_and_ln_c3_size_vs_library(dev):
    """Config-3 size: T = 655,360 token rows.  The 512 -> 2048 Linear (forward, dX, dW over the token
    splits) and the 2048-wide fused LayerNorm-GELU(-Dropout), forward and backward, against the library on the device.  Y / E are
    5.4 GB each (byte offsets beyond 2^32); sampled rows cover the whole range and the tail."""
    import torch.nn.functional as F
    from madeleine_amd import functional as MF
    T, N, K = 160 * 4096, 2048, 512
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(T, K, device=dev, generator=g).requires_grad_()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.04).requires_grad_()
    dy = torch.randn(T, N, device=dev, generator=g)
    y = MF.linear(x, W)
    gx, gW = torch.autograd.grad(y, (x, W), dy)
    yr = F.linear(x, W)
    rx, rW = torch.autograd.grad(yr, (x, W), dy)
    for rows in (slice(0, T, 997), slice(T - 300, T)):
        assert rel_err(y[rows], yr[rows]) < 1e-5 and rel_err(gx[rows], rx[rows]) < 1e-5
    assert rel_err(gW, rW) < 1e-4      # 655,360-term fp32 sums in two different orders
    del gx, rx, x, dy

    # fused LN-GELU on the 2048-wide rows (eval: dropout off), then the in-kernel dropout's consistency on the tail rows
    y = y.detach()
    gam = (1 + 0.2 * torch.randn(N, device=dev, generator=g)).requires_grad_()
    bet = (0.3 * torch.randn(N, device=dev, generator=g)).requires_grad_()
    lb = (0.1 * torch.randn(N, device=dev, generator=g)).requires_grad_()
    dz = torch.randn(T, N, device=dev, generator=g)
    yd = y.clone().requires_grad_()
    out = MF.ln_gelu_drop(yd, gam, bet, 1e-5, 0.0, 0, None, lb)
    o_g = torch.autograd.grad(out, (yd, gam, bet, lb), dz)
    ref_in = yr.detach().clone().requires_grad_()
    g2, b2, lb2 = (v.detach().clone().requires_grad_() for v in (gam, bet, lb))
    ref = F.gelu(F.layer_norm(ref_in + lb2, (N,), g2, b2, 1e-5))
    r_g = torch.autograd.grad(ref, (ref_in, g2, b2, lb2), dz)
    for rows in (slice(0, T, 997), slice(T - 300, T)):
        assert rel_err(out[rows], ref[rows]) < 1e-5 and max_rel(out[rows], ref[rows]) < TOL
        assert rel_err(o_g[0][rows], r_g[0][rows]) < 1e-4
    for a, b in zip(o_g[1:], r_g[1:]):
        assert rel_err(a, b) < 1e-4
    out_d = MF.ln_gelu_drop(yd, gam, bet, 1e-5, 0.1, 4242, None, lb).detach()
    tail_d, tail_0 = out_d[T - 4096:], out[T - 4096:].detach()
    dropped = (tail_d == 0) & (tail_0 != 0)
    assert abs(float(dropped.float().mean()) - 0.1) < 0.005
    assert rel_err(tail_d[~dropped], (tail_0 / 0.9)[~dropped]) < 1e-6


@pytest.mark.parametrize("din,hid,ncls,N", [(1024, 256, 1, 700), (1024, 256, 1, 40), (768, 512, 2, 300)])
def test_batched_abmil_other_geometries_vs_oracle(dev, din, hid, ncls, N):
    """BatchedABMIL outside the 512 / 512 / 1 geometry MADELEINE hard-wires -- first of all the reference class's own defaults
    input_dim = 1024, hidden_dim = 256 (madeleine/models/abmil.py:10; VERDICT round 2, missing #4): values, the softmax over the patch
    axis and every gradient against the oracle (abmil.py:41-68), train mode with injected dropout masks."""
    from madeleine_amd import BatchedABMIL
    from oracle import recipe
    B = 2
    m = BatchedABMIL(input_dim=din, hidden_dim=hid, dropout=True, n_classes=ncls).to(dev).train()
    sd = {k: torch.from_numpy(v) for k, v in recipe.state_dict_recipe({k: tuple(v.shape) for k, v in m.state_dict().items()}, "abg").items()}
    m.load_state_dict(sd)
    x = t((B, N, din), f"abg:x{din}{N}")
    ka = torch.from_numpy(recipe.bernoulli((B, N, hid), "abg:ka", 0.75))
    kb = torch.from_numpy(recipe.bernoulli((B, N, hid), "abg:kb", 0.75))
    gout = t((B, N, ncls), "abg:g")
    ref_p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    xr = x.clone().requires_grad_()
    raw_ref = R.gate_scores(xr, ref_p["attention_a.0.weight"], ref_p["attention_a.0.bias"], ref_p["attention_b.0.weight"],
                            ref_p["attention_b.0.bias"], ref_p["attention_c.weight"], ref_p["attention_c.bias"], ka, kb)
    att_ref = torch.softmax(raw_ref, dim=1)
    ((att_ref * gout).sum() + 0.3 * (raw_ref * gout).sum()).backward()   # (the raw term: softmax alone is shift-invariant in bc)
    m._injected_keep = (ka.to(dev), kb.to(dev))
    xd = x.to(dev).requires_grad_()
    att, raw = m(xd, return_raw_attention=True)
    ((att * gout.to(dev)).sum() + 0.3 * (raw * gout.to(dev)).sum()).backward()
    assert rel_err(raw, raw_ref) < 1e-5 and rel_err(att, att_ref) < 1e-5
    assert rel_err(xd.grad, xr.grad) < 1e-4
    for k, p in m.named_parameters():
        assert rel_err(p.grad, ref_p[k].grad) < 1e-4, k


def test_dropout_masks_do_not_repeat_beyond_2_pow_24_elements(dev):
    """the counter hash must spread 32-bit counters.  With round 4's 24-bit multiplies element idx and
    element idx ^ (d << 24 | d << 8) always shared their keep decision -- the masks of token rows 2^24 / (H * 512) = 8192 tokens apart
    repeated.  The exported gate masks ([T, H, 512] -> flat index (t * H + c) * 512 + e) of T = 3 * 8192 + 64 tokens: the keep rate is
    exact, those pairs are no longer tied (a two-multiply finaliser keeps a residual correlation on such structured differences --
    tools/rng_quality.py: worst 0.22 over all 1- / 2-bit and (d << 24 | d << 8) differences, lowbias32 0.27 -- bounded here at 0.3),
    and whole token rows 8192 tokens apart agree on no more elements than independent masks do."""
    from tests.test_bench_path_gpu import _exported_masks
    T, H, p = 3 * 8192 + 64, 4, 0.25
    ka, kb = _exported_masks(dev, T, H, p, 987654321)
    for m in (ka, kb):
        flat = m.reshape(-1).float()
        n = flat.numel()
        assert abs(float(flat.mean()) - 0.75) < 5.0 * (0.75 * 0.25 / n) ** 0.5
        idx = torch.arange(0, n - (3 << 24) - 4096, 7, device=dev)
        z = flat - flat.mean()
        for d in (1, 2, 3):
            partner = idx ^ ((d << 24) | (d << 8))
            ok = partner < n
            c = float((z[idx[ok]] * z[partner[ok]]).mean() / z.var())
            assert abs(c) < 0.3, (d, c)      # round 4: exactly 1.0
        # whole 2048-element rows (one token, all heads) 8192 tokens apart
        rows = m.reshape(T, H * 512)
        assert float((rows[:8192] == rows[8192:16384]).float().mean()) < 0.64      # independent: 0.75^2 + 0.25^2 = 0.625
