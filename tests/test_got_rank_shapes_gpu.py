"""GOT at the shapes the 8-GPU configurations give ONE rank (reference madeleine/utils/loss.py:179-193,236-248,278-302):

  (a) four problems x k = 32 local cases x n = 256 tokens in ONE batched launch sequence (BASELINE configs[3] with every stain present, and
      every rank of configs[4]) -- the n in (192, 256] class of the one-pass IPOT reverse sweep (csrc/got_impl.inc, LDS-resident accumulator
      rows) -- and a mixed batch n = 256 / 200 / 130 / 64, against the fp64 oracle: both distances and the token gradients, twice for
      bit-reproducibility, batched == single-problem for the top size class;
  (b) the config-5 rank shape through calculate_losses_dp: ragged bags with stain tokens -> the first 256 tokens of every bag -> k = 256,
      n = 256, with the oracle fed the IDENTICAL token / slide embeddings (an end-to-end tolerance then separates a
      regression of the sweep from amplified encoder rounding);
  (c) bench.emulated_rank_loss -- the helper behind bench.py's c4 / c5 rank-emulation legs -- against the oracle evaluated on the
      concatenated global batch (8 emulated ranks x 2 local cases).
Test infrastructure: the oracle is the checker, the HIP path (through the C ABI) is what is checked.
"""
import os
from types import SimpleNamespace

import pytest
import torch

from oracle import restatement as R
from tests._util import MODS5, rel_err, t

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _problems(shape, tag):
    probs, refs = [], []
    for s, (k, n) in enumerate(shape):
        v = t((k, n, 128), f"got:{tag}:v{s}")
        q = t((k, n, 128), f"got:{tag}:q{s}") + 0.7 * v
        v64, q64 = v.double().requires_grad_(), q.double().requires_grad_()
        ref = R.got(v64, q64, subsample=None)
        ref.backward()
        probs.append((v, q))
        refs.append((float(ref.detach()), v64.grad, q64.grad))
    return probs, refs


def _run_batched(dev, probs_cpu):
    from madeleine_amd import distributed as DP
    from madeleine_amd import functional as MF
    probs = [(v.to(dev).requires_grad_(), q.to(dev).requires_grad_()) for v, q in probs_cpu]
    ext = DP.got_local_extrema([(a.detach(), b.detach()) for a, b in probs], MF.HipGotImpl)
    outs = DP.got_multi(probs, MF.HipGotImpl, None, extrema=ext)            # [S, 2]
    (outs[:, 0] + outs[:, 1]).sum().backward()
    torch.cuda.synchronize()
    return outs.detach().clone(), [(a.grad.clone(), b.grad.clone()) for a, b in probs]


SHAPES = {
    # (cases this rank owns, n = min(k_global, 256)) per stain
    "all_present_4x32x256": [(32, 256)] * 4,      # configs[3] with every stain on every case; every rank of configs[4]
    "mixed_256_200_130_64": [(32, 256), (12, 200), (9, 130), (6, 64)],
}


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_got_multi_n256_class_vs_fp64_oracle(dev, name):
    from madeleine_amd import functional as MF
    shape = SHAPES[name]
    probs_cpu, refs = _problems(shape, name)
    assert MF.HipGotImpl.can_batch([(v.to(dev), q.to(dev)) for v, q in probs_cpu])   # ONE batched launch sequence, not the stream fan-out
    o1, g1 = _run_batched(dev, probs_cpu)
    o2, g2 = _run_batched(dev, probs_cpu)
    assert torch.equal(o1, o2)
    for (a1, b1), (a2, b2) in zip(g1, g2):
        assert torch.equal(a1, a2) and torch.equal(b1, b2)
    for s, (ref, dv, dq) in enumerate(refs):
        got = float(o1[s].sum())
        ev, eq = rel_err(g1[s][0], dv), rel_err(g1[s][1], dq)
        if os.environ.get("MDL_TEST_VERBOSE"):
            print("GOTERR %s problem %d (k=%d n=%d): value rel %.2e  dV rel %.2e  dQ rel %.2e" % (name, s, *shape[s], abs(got - ref) / abs(ref), ev, eq))
        # north_star's bar is 1e-3; measured on MI355X (profiles/r06_got_rank_shape_errors.txt): values <= 2.6e-6, gradients <= 2e-7 --
        # held at ~10x that, so a regression of the sweep shows long before the 1e-3 bar
        assert abs(got - ref) < 3e-5 * abs(ref), (s, got, ref)
        assert ev < 3e-6 and eq < 3e-6, (s, ev, eq)
    # the same problems through the single-problem entry points: the batched launches run every problem on the kernels of the LARGEST
    # problem's size class, so problems of that class give the same bits either way
    top = max(n for _, n in shape)
    for s, (v, q) in enumerate(probs_cpu):
        vd, qd = v.to(dev).requires_grad_(), q.to(dev).requires_grad_()
        o = MF.got(vd, qd)
        (o[0] + o[1]).backward()
        if shape[s][1] == top:
            assert torch.equal(o, o1[s]) and torch.equal(vd.grad, g1[s][0]) and torch.equal(qd.grad, g1[s][1]), s
        else:
            assert rel_err(o, o1[s]) < 1e-5 and rel_err(vd.grad, g1[s][0]) < 1e-4 and rel_err(qd.grad, g1[s][1]) < 1e-4, s


def test_c5_rank_shape_through_calculate_losses_dp_identical_tokens(dev):
    """One stain, 256 participating cases, ragged bags of 256..400 patches with stain-encoding tokens (BASELINE configs[4] geometry at
    d = 96): MADELEINE.forward_ragged keeps the first 256 tokens of every bag, calculate_losses_dp slices n = min(k, 256) = 256 of them
    per case into ONE GOT problem of the top size class (k = 256: eight times a rank's 32 cases) and adds the global InfoNCE.  The oracle
    receives the token and slide embeddings the HIP encoder produced (detached, fp64): loss and d loss / d embeddings are then a statement
    about the loss kernels and the gather / scatter glue alone."""
    from madeleine_amd import InfoNCE
    from madeleine_amd import distributed as DP
    from madeleine_amd import functional as MF
    from tests.test_model_gpu import build
    B, M, D = 256, 2, 96
    mods = MODS5[:M]
    model = build(mods, D, "c5rank", dev, stain_encoding=True).eval()
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(256, 401, (B, M), generator=g)
    bags = [[t((int(lens[b, m]), D), f"c5rank:f{b}:{m}") for m in range(M)] for b in range(B)]
    labels = torch.ones(B, M)
    largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.5)
    T_ = 0.1
    embs, toks = model.forward_ragged(bags, dev)
    assert toks[mods[1]].shape == (B, 256, 128)
    leaves = {"e_he": embs["HE"], "e_st": embs[mods[1]], "t_he": toks["HE"], "t_st": toks[mods[1]]}
    for v in leaves.values():
        v.retain_grad()
    loss, flag = DP.calculate_losses_dp(mods[1:], InfoNCE(temperature=T_), MF.HipGotImpl, embs, toks, labels[:, 1:], largs)
    assert flag
    loss.backward()
    torch.cuda.synchronize()
    ref_in = {k: v.detach().double().cpu().requires_grad_() for k, v in leaves.items()}
    ref = R.info_nce(ref_in["e_he"][:, 0, :, 0], ref_in["e_st"][:, 0, :], T_, True) \
        + 0.5 * R.got(ref_in["t_he"][:, :, :, 0], ref_in["t_st"], subsample=None)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-5 * abs(float(ref.detach())), (float(loss.detach()), float(ref.detach()))
    for k, v in leaves.items():
        err = rel_err(v.grad, ref_in[k].grad)
        if os.environ.get("MDL_TEST_VERBOSE"):
            print("GOTERR c5 rank shape (k=256 n=256, identical tokens): d loss / d %s rel %.2e; loss rel %.2e" % (
                k, err, abs(float(loss.detach()) - float(ref.detach())) / abs(float(ref.detach()))))
        assert err < 3e-5, (k, err)     # measured on MI355X: 3.8e-7 .. 2.4e-6 (profiles/r06_got_rank_shape_errors.txt)
    assert float(leaves["t_st"].grad.abs().sum()) > 0


def test_bench_rank_emulation_equals_oracle_on_the_concatenated_global_batch(dev):
    """bench.emulated_rank_loss (the c4 / c5 rank-emulation legs): 8 emulated ranks x B_l = 2 local cases, 3 stains with a mixed presence
    pattern.  (i) Global batch = this rank's 2 cases + 7 emulated copies: slide embeddings perturbed by the helper's noise, tokens and
    labels tiled (so every rank's GOT share -- and the batch extrema -- equal the local ones).  The oracle's calculate_losses on that
    concatenated 16-case batch (InfoNCE over the global rows; GOT at n = min(k_global, 256) tokens over all 16 cases, trainer.py:20-77 +
    loss.py:278-302) must give the emulated rank's loss: global InfoNCE + 8 x the local GOT sum.  (ii) Labels that are NOT tiled (a stain
    whose participating cases all live on other ranks, another whose global count differs from 8 x the local one): value and gradients
    w.r.t. the local embeddings against an fp64 restatement of the rank's formula."""
    import bench as BN
    from madeleine_amd import InfoNCE
    from madeleine_amd import distributed as DP
    from madeleine_amd import functional as MF
    W, Bl, M, N = 8, 2, 4, 40
    mods = MODS5[:M]
    labels = torch.tensor([[1., 1., 1., 0.], [1., 1., 0., 0.]])       # stain 1: both cases; stain 2: one case; stain 3: no local case
    T_ = 0.1
    largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.5)
    noise = {m: 0.05 * t(((W - 1) * Bl, 1, 512), f"emu:n:{m}").to(dev) for m in mods}
    g_loss = lambda a, b, symmetric=False: R.info_nce(a, b, T_, symmetric)     # noqa: E731
    l_loss = lambda a, b, subsample=None: R.got(a, b, subsample)               # noqa: E731

    def hip(lab_g):
        e = {m: (t((Bl, 1, 512), f"emu:e:{m}")).to(dev).requires_grad_() for m in mods}
        tk = {m: (t((Bl, N, 128), f"emu:t:{m}") + (0.7 * t((Bl, N, 128), "emu:t:HE") if m != "HE" else 0)).to(dev).requires_grad_()
              for m in mods}
        embs = {m: (e[m].unsqueeze(3).expand(-1, -1, -1, M - 1) if m == "HE" else e[m]) for m in mods}
        toks = {m: (tk[m].unsqueeze(3).expand(-1, -1, -1, M - 1) if m == "HE" else tk[m]) for m in mods}
        loss = BN.emulated_rank_loss(DP, MF, InfoNCE(temperature=T_), largs, mods, embs, toks, labels, lab_g, noise, W)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach(), e, tk

    def global_batch(e, tk):
        e64 = {m: e[m].detach().double().cpu().requires_grad_() for m in mods}
        t64 = {m: tk[m].detach().double().cpu().requires_grad_() for m in mods}
        eg, tg = {}, {}
        for m in mods:
            full = torch.cat([e64[m], e64[m].detach().repeat(W - 1, 1, 1) + noise[m].double().cpu()])
            tfull = torch.cat([t64[m]] + [t64[m].detach()] * (W - 1))
            eg[m] = full.unsqueeze(3).repeat(1, 1, 1, M - 1) if m == "HE" else full
            tg[m] = tfull.unsqueeze(3).repeat(1, 1, 1, M - 1) if m == "HE" else tfull
        return e64, t64, eg, tg

    def rank_formula(lab_g, e64, t64, eg):
        k_g = [int(lab_g[:, s].sum()) for s in range(1, M)]
        ref = 0.0
        for s, stain in enumerate(mods[1:]):
            mask_g = lab_g[:, 1 + s].bool()
            if k_g[s] <= 1:
                continue
            ref = ref + R.info_nce(eg["HE"][:, 0, :, s][mask_g], eg[stain][:, 0, :][mask_g], T_, True)
            mask_l = labels[:, 1 + s].bool()
            if int(mask_l.sum()) > 0:
                n = min(k_g[s], 256)
                ref = ref + W * 0.5 * R.got(t64["HE"][mask_l][:, :n], t64[stain][mask_l][:, :n], subsample=None)
        return ref

    # (i) tiled labels: the oracle on the concatenated global batch
    lab_g = labels.repeat(W, 1)                                       # cases per stain [16, 8, 0]
    loss, e, tk = hip(lab_g)
    e64, t64, eg, tg = global_batch(e, tk)
    ref_global, flag = R.calculate_losses(mods[1:], g_loss, l_loss, None, eg, tg, lab_g[:, 1:], True, 0.5)
    assert flag
    assert abs(float(loss) - float(ref_global.detach())) < 1e-4 * abs(float(ref_global.detach())), (float(loss), float(ref_global.detach()))
    ref = rank_formula(lab_g, e64, t64, eg)
    assert abs(float(ref.detach()) - float(ref_global.detach())) < 1e-9 * abs(float(ref_global.detach()))

    # (ii) labels that differ between the ranks
    lab_g = labels.repeat(W, 1)
    lab_g[5, 3] = lab_g[8, 3] = lab_g[11, 3] = 1                      # stain 3: three cases, all on other ranks
    lab_g[4, 2] = 0                                                   # stain 2: 7 cases globally, one of them here
    loss, e, tk = hip(lab_g)
    e64, t64, eg, tg = global_batch(e, tk)
    ref = rank_formula(lab_g, e64, t64, eg)
    assert abs(float(loss) - float(ref.detach())) < 1e-4 * abs(float(ref.detach())), (float(loss), float(ref.detach()))
    ref.backward()
    for m in mods:
        assert rel_err(e[m].grad, e64[m].grad) < TOL, m
        if t64[m].grad is not None and float(t64[m].grad.norm()) > 0:
            assert rel_err(tk[m].grad, t64[m].grad) < TOL, m
        else:
            assert tk[m].grad is None or float(tk[m].grad.abs().max()) == 0.0, m


def test_split_sweeps_equal_one_workgroup_sweeps_and_do_not_depend_on_timing(dev):
    """Round 6: the IPOT sweeps of the n <= 192 / 256 classes run as TWO workgroups per case and branch that exchange their column sums
    once per iteration (csrc/got_impl.inc, Xch).  (i) Same numbers as the one-workgroup sweeps (MADELEINE_GOT_NOSPLIT=1) up to the
    re-association of the column sums; (ii) bit-identical from run to run while another stream keeps the memory system and the compute
    units busy (the exchange is value-deterministic: a stale or torn granule would change bits); (iii) no exchange timed out."""
    from madeleine_amd import functional as MF
    for name, shape in (("4x32x256", [(32, 256)] * 4), ("3x20x180", [(20, 180)] * 3)):
        probs_cpu, _ = _problems(shape, "split:" + name)
        o1, g1 = _run_batched(dev, probs_cpu)
        os.environ["MADELEINE_GOT_NOSPLIT"] = "1"
        try:
            o0, g0 = _run_batched(dev, probs_cpu)
        finally:
            del os.environ["MADELEINE_GOT_NOSPLIT"]
        assert rel_err(o1, o0) < 1e-6, (name, rel_err(o1, o0))
        for (a1, b1), (a0, b0) in zip(g1, g0):
            assert rel_err(a1, a0) < 2e-6 and rel_err(b1, b0) < 2e-6, (name, rel_err(a1, a0), rel_err(b1, b0))
        assert not torch.equal(o1, o0) or not all(torch.equal(a1, a0) for (a1, _), (a0, _) in zip(g1, g0))   # the switch did switch paths
        # (ii) under load from a second stream: 1-GiB copies that occupy compute units and HBM while the sweeps exchange
        src = torch.empty(256 * 2 ** 20, device=dev)
        dst = torch.empty_like(src)
        side = torch.cuda.Stream()
        for rep in range(4):
            with torch.cuda.stream(side):
                for _ in range(12):
                    dst.copy_(src)
            o2, g2 = _run_batched(dev, probs_cpu)
            side.synchronize()
            assert torch.equal(o2, o1), (name, rep)
            for (a2, b2), (a1, b1) in zip(g2, g1):
                assert torch.equal(a2, a1) and torch.equal(b2, b1), (name, rep)
    # (iii) the time-out flag of a workspace after a forward and a backward pass at k = 32, n = 256
    v, q = _problems([(32, 256)], "split:flag")[0][0]
    v, q = v.to(dev), q.to(dev)
    out, state = MF.HipGotImpl.forward(v, q, MF.got_extrema(v, q))
    assert MF.got_exchange_timeouts(state[2]) == 0.0
    MF.HipGotImpl.backward_begin(state, torch.ones(2, device=dev))
    assert MF.got_exchange_timeouts(state[2]) == 0.0 and torch.isfinite(out).all()


@pytest.mark.parametrize("k,n", [(6, 256), (5, 150), (4, 100), (3, 40), (2, 300)])
def test_void_pass_comes_back_as_nan_not_as_numbers(dev, k, n):
    """A split sweep that gives up on its partner (csrc/got_impl.inc, Xch: e.g. two processes sharing the GPU, each assuming it owns every
    compute unit) raises the workspace's time-out flag.  Nothing polls that flag in a training loop, so the pass itself must say it is
    void: with the flag raised the backward returns NaN token gradients (and a forward NaN distances: got_sum_kernel reads the same
    word).  Every size class defines the flag (cleared by the extrema kernel), so regular passes stay finite in all of them."""
    from madeleine_amd import functional as MF
    v = t((k, n, 128), "void:v%d:%d" % (k, n))
    q = (t((k, n, 128), "void:q%d:%d" % (k, n)) + 0.7 * v).to(dev)
    v = v.to(dev)
    out, state = MF.HipGotImpl.forward(v, q, MF.got_extrema(v, q))
    assert torch.isfinite(out).all() and MF.got_exchange_timeouts(state[2]) == 0.0
    dmm = MF.HipGotImpl.backward_begin(state, torch.ones(2, device=dev))
    dV, dQ = MF.HipGotImpl.backward_finish(state, dmm)
    assert torch.isfinite(dV).all() and torch.isfinite(dQ).all() and float(dV.abs().sum()) > 0
    # the same backward on a workspace whose flag is up
    out2, state2 = MF.HipGotImpl.forward(v, q, MF.got_extrema(v, q))
    assert torch.equal(out2, out)
    MF._got_set_exchange_timeout(state2[2])
    dmm2 = MF.HipGotImpl.backward_begin(state2, torch.ones(2, device=dev))
    dV2, dQ2 = MF.HipGotImpl.backward_finish(state2, dmm2)
    assert torch.isnan(dV2).all() and torch.isnan(dQ2).all()
