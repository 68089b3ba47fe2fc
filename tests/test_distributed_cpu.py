"""World-size-2 gloo test (CPU) of the data-parallel decomposition in madeleine_amd/distributed.py:
W-rank sharded loss / parameter gradients == single-process global-batch loss / gradients, which is the
semantics nn.DataParallel gives the reference.  The numeric kernels are replaced by the
CPU oracle through the same interfaces (loss callable, GOT impl), so only the host/collective logic is under test."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import restatement as R
from tests._util import MODS5, recipe_params, t


class OracleGotImpl:
    """CPU stand-in for madeleine_amd.functional.HipGotImpl (same four-stage contract), built on the oracle."""

    @staticmethod
    def extrema(V, Q):
        with torch.no_grad():
            return R.got_extrema(V, Q)

    @staticmethod
    def forward(V, Q, minmax):
        v, q, mm = V.detach().requires_grad_(), Q.detach().requires_grad_(), minmax.detach().requires_grad_()
        with torch.enable_grad():
            out = R.got_parts(v, q, mm)
        return out.detach(), {"v": v, "q": q, "mm": mm, "out": out}

    @staticmethod
    def backward_begin(st, d_out):
        gv, gq, gmm = torch.autograd.grad(st["out"], [st["v"], st["q"], st["mm"]], d_out)
        st["gv"], st["gq"] = gv, gq
        return gmm

    @staticmethod
    def backward_finish(st, dmm_total):
        v, q = st["v"].detach().requires_grad_(), st["q"].detach().requires_grad_()
        with torch.enable_grad():
            e = R.got_extrema(v, q)
        owned = (e.detach() == st["mm"].detach()).to(e.dtype)        # extrema attained on this rank
        ev, eq = torch.autograd.grad(e, [v, q], dmm_total * owned)
        return st["gv"] + ev, st["gq"] + eq


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


N, D = 14, 64
CASES = {
    # benign sharding: every participating stain has cases on both ranks.  HER2: k_global 4 (ranks 2/2); PGR: 5 (2/3); KI67: 4 (1/3)
    "benign": torch.tensor([[1, 1, 1, 0], [1, 1, 0, 1], [1, 0, 1, 1], [1, 1, 1, 1], [1, 1, 1, 0], [1, 0, 1, 1]], dtype=torch.float32),
    # degenerate sharding (the reference gets these for free from the gathered batch, trainer.py:25-26,71-75):
    #   W = 4 (2 cases per rank): rank 1 = cases 2, 3 is H&E-only; rank 1 owns no case of any participating stain; rank 0 has no ER case
    #   W = 8 (1 case per rank) : most ranks own zero cases of most stains
    #   KI67: k_global = 1 (case 7) -> skipped (trainer.py:28); ER: k_global = 2, one case on each of two ranks (4 | 7)
    "degenerate": torch.tensor([[1, 1, 1, 0, 0],
                                [1, 1, 0, 0, 0],
                                [1, 0, 0, 0, 0],
                                [1, 0, 0, 0, 0],
                                [1, 1, 1, 0, 1],
                                [1, 1, 0, 0, 0],
                                [1, 1, 1, 0, 0],
                                [1, 0, 1, 1, 1]], dtype=torch.float32),
    # nothing but H&E anywhere (or single cases): the (-1, False) sentinel on every rank (trainer.py:72-75)
    "he_only": torch.tensor([[1, 0, 0, 0], [1, 1, 0, 0], [1, 0, 0, 0], [1, 0, 0, 1]], dtype=torch.float32),
}


def _params64(M):
    """fp64 leaves: the decomposition is exact in real arithmetic, so fp64 pins the LOGIC to ~1e-10 (in fp32 the
    GW fixed point amplifies summation-order noise to ~1e-3 on some gradients, which would hide logic errors)."""
    return {k: v.double().requires_grad_() for k, v in recipe_params(M, D, "wdp").items()}


def _single_process(use_got, case="benign"):
    LABELS = CASES[case]
    B, M = LABELS.shape
    mods = MODS5[:M]
    sd = _params64(M)
    feats = t((B, M, N, D), "dp:feats").double()
    nce = lambda a, b, symmetric=False: R.info_nce(a, b, 0.001, symmetric)  # noqa: E731
    embs, toks = R.madeleine_forward_train(feats, sd, mods)
    # identity token permutation: the value does not depend on the randperm order
    loc = (lambda a, b, subsample=None: R.got(a, b, subsample, perm=torch.arange(a.shape[0]))) if use_got else None
    loss, flag = R.calculate_losses(mods[1:], nce, loc, None, embs, toks, LABELS[:, 1:], True, 0.7)
    if not flag:
        return loss, None
    loss.backward()
    return float(loss), {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}


def _worker(rank, world, port, use_got, ret, case="benign"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2 if world <= 2 else 1)
    LABELS = CASES[case]
    B, M = LABELS.shape
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madeleine_amd import distributed as DP
        mods = MODS5[:M]
        sd = _params64(M)
        Bl = B // world
        sl = slice(rank * Bl, (rank + 1) * Bl)
        feats = t((B, M, N, D), "dp:feats").double()[sl]
        nce = lambda query, positive_key, symmetric=False: R.info_nce(query, positive_key, 0.001, symmetric)  # noqa: E731
        args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.7)
        embs, toks = R.madeleine_forward_train(feats, sd, mods)
        loss, flag = DP.calculate_losses_dp(mods[1:], nce, OracleGotImpl if use_got else None, embs, toks,
                                            LABELS[sl, 1:], args, use_local_loss=use_got)
        if not flag:       # sentinel: every rank must reach the same verdict without touching a collective it would hang in
            ret["r%d" % rank] = (loss, bool(flag))
            return
        loss.backward()
        # the gradient mean over ranks bench.py performs for N > 1: distributed.FlatGradSync (one packed all-reduce); the parameters
        # outside the step's graph (token_projector without the local loss) are excluded on every rank alike
        sync = DP.FlatGradSync(sd.items(), use_local_loss=use_got)
        assert all(v.grad is not None for v in sync.params)
        sync.all_reduce_mean()
        assert all(v.grad.data_ptr() == w.data_ptr() for v, w in zip(sync.params, sync.views))
        grads = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
        # true loss value = replicated global part + sum over ranks of the local part (undo the W scaling)
        with torch.no_grad():
            embs_g = DP.gather_slide_embeddings({k: v.detach() for k, v in embs.items()}, mods)
            labels_g = DP.all_gather_labels(LABELS[sl, 1:], torch.device("cpu"))
            lg, _ = R.calculate_losses(mods[1:], lambda a, b, symmetric=False: R.info_nce(a, b, 0.001, symmetric), None, None,
                                       embs_g, None, labels_g, True, 0.7)
        local_part = (loss.detach() - lg) / world
        dist.all_reduce(local_part)
        if rank == 0:
            ret["loss"] = float(lg + local_part)
            ret["grads"] = {k: v.numpy() for k, v in grads.items()}
            ret["flag"] = bool(flag)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_got", [False, True])
def test_two_rank_gloo_equals_global_batch(use_got):
    ref_loss, ref_grads = _single_process(use_got)
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, use_got, ret), nprocs=2, join=True)
    assert ret["flag"]
    assert abs(ret["loss"] - ref_loss) < 1e-9 * abs(ref_loss), (ret["loss"], ref_loss)
    top = max(float(g.norm()) for g in ref_grads.values())
    for k, g in ref_grads.items():
        got = torch.from_numpy(ret["grads"][k])
        err = float((got - g).norm())
        assert err <= 1e-8 * float(g.norm()) + 1e-10 * top, (k, err, float(g.norm()))


def _check_against_global_batch(world, use_got, case):
    ref_loss, ref_grads = _single_process(use_got, case)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), use_got, ret, case), nprocs=world, join=True)
    assert ret["flag"]
    assert abs(ret["loss"] - ref_loss) < 1e-9 * abs(ref_loss), (ret["loss"], ref_loss)
    top = max(float(g.norm()) for g in ref_grads.values())
    for k, g in ref_grads.items():
        got = torch.from_numpy(ret["grads"][k])
        err = float((got - g).norm())
        assert err <= 1e-8 * float(g.norm()) + 1e-10 * top, (world, k, err, float(g.norm()))


@pytest.mark.parametrize("use_got", [False, True])
@pytest.mark.parametrize("world", [4, 8])
def test_degenerate_sharding_equals_global_batch(world, use_got):
    """W = 4 and W = 8 on a label matrix whose shards are as uneven as a real batch allows (CASES['degenerate']): a rank that is
    H&E-only while the global batch is not, ranks with zero local cases of a participating stain (empty GOT problems that still take
    part in the [S,6] all-reduce), a stain with k_global = 1 (skipped on every rank) and one with k_global = 2 split over two ranks
    (each rank's GOT problem has ONE case and n = 2 tokens).  The W-rank loss and FlatGradSync-averaged parameter gradients must equal
    the single-process global-batch ones (reference semantics: nn.DataParallel gathers before the loss, setup_components.py:185-187;
    trainer.py:25-26,71-75; loss.py:282,289-292)."""
    _check_against_global_batch(world, use_got, "degenerate")


@pytest.mark.parametrize("world", [2, 4])
def test_he_only_global_batch_is_the_sentinel_on_every_rank(world):
    """No stain with more than one case in the GLOBAL batch: every rank returns (-1, False) (trainer.py:72-75) -- and must do so
    without leaving a peer inside a collective (the ranks that do hold a stain case must not start the GOT extrema exchange)."""
    ref_loss, ref_grads = _single_process(True, "he_only")
    assert ref_loss == -1 and ref_grads is None
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), True, ret, "he_only"), nprocs=world, join=True)
    for r in range(world):
        loss, flag = ret["r%d" % r]
        assert loss == -1 and flag is False


def test_got_parts_matches_got():
    """the DP building block reduces to the pinned reference restatement when thresholds are batch-local"""
    v = t((4, 9, 128), "dp:gv")
    q = t((4, 9, 128), "dp:gq") + 0.6 * v
    a = R.got(v, q)
    b = R.got_parts(v, q).sum()
    assert abs(float(a) - float(b)) < 1e-6 * abs(float(a))
    # halves with the global extrema add up to the whole
    ex = R.got_extrema(v, q)
    c = R.got_parts(v[:2], q[:2], ex) + R.got_parts(v[2:], q[2:], ex)
    assert abs(float(c.sum()) - float(a)) < 1e-5 * abs(float(a))


def _missing_grad_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madeleine_amd import distributed as DP
        a = torch.nn.Parameter(torch.ones(3))
        b = torch.nn.Parameter(torch.ones(2))
        sync = DP.FlatGradSync([("a", a), ("b", b)])
        a.grad = torch.full((3,), float(rank + 1))
        b.grad = torch.full((2,), 2.0 * (rank + 1))
        sync.all_reduce_mean()                       # a complete step: the mean, and the flag slot stays out of the gradients
        ok = bool(torch.allclose(a.grad, torch.full((3,), 1.5)) and torch.allclose(b.grad, torch.full((2,), 3.0)))
        a.grad = torch.ones(3)
        b.grad = None if rank == 1 else torch.ones(2)      # only rank 1 misses a gradient
        try:
            sync.all_reduce_mean()
            raised = ""
        except RuntimeError as e:
            raised = str(e)
        dist.barrier()                                # nobody is stuck inside a collective
        ret["r%d" % rank] = (ok, raised)
    finally:
        dist.destroy_process_group()


def test_flat_grad_sync_missing_gradient_raises_on_every_rank():
    """A packed parameter without a gradient on ONE rank: every rank raises in the same step, after the collective (a rank-local raise
    in front of it would leave the peers inside the all-reduce until the backend's timeout)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_missing_grad_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    ok0, msg0 = ret["r0"]
    ok1, msg1 = ret["r1"]
    assert ok0 and ok1
    assert "other rank" in msg0, msg0
    assert "no gradient" in msg1 and "b" in msg1, msg1
