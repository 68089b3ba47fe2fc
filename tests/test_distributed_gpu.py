"""Two-rank data-parallel step on the GPU with the real HIP kernels ("make the multi-GPU path provably
ready"): encoder + calculate_losses_dp (one packed all-gather: slide embeddings | presence mask | GOT extrema; [S,6]
all-reduce in backward) + the gradient mean DDP performs, against the single-process global-batch HIP result -- the
semantics nn.DataParallel gives the reference (setup_components.py:185-187).

  backend "nccl" : RCCL over xGMI, one rank per GPU -- needs >= 2 GPUs (skipped on a 1-GPU box);
  backend "gloo" : both ranks share cuda:0 and the collectives are host-staged (distributed._host_staged) -- the same
                   host logic, packing, autograd nodes and kernels, runnable on one GPU.
Tolerance: the two sides run the same kernels on different batch partitions, so sums are re-associated: 1e-4 relative on the
loss and on every parameter gradient (GOT's fixed point amplifies summation-order noise, cf. tests/test_distributed_cpu.py
where the fp64 oracle pins the decomposition itself to 1e-9)."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests._util import MODS5, t

pytestmark = pytest.mark.gpu

B, M, N, D = 8, 4, 300, 64
LABELS = torch.tensor([[1, 1, 1, 0], [1, 1, 0, 1], [1, 0, 1, 1], [1, 1, 1, 1],
                       [1, 1, 1, 0], [1, 0, 1, 1], [1, 1, 1, 1], [1, 0, 0, 1]], dtype=torch.float32)
# the degenerate sharding of tests/test_distributed_cpu.py (CASES["degenerate"]): at W = 4 rank 1 (cases 2, 3) is H&E-only, ranks own zero
# cases of participating stains, KI67 has k_global = 1 (skipped), ER has k_global = 2 with one case on each of two ranks
LABELS_DEGENERATE = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 0, 0, 0], [1, 0, 0, 0, 0], [1, 0, 0, 0, 0],
                                  [1, 1, 1, 0, 1], [1, 1, 0, 0, 0], [1, 1, 1, 0, 0], [1, 0, 1, 1, 1]], dtype=torch.float32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build_model(dev, n_mod=M):
    from tests.test_model_gpu import build
    return build(MODS5[:n_mod], D, "wdp", dev).eval()   # eval: dropout off, gradients still flow (parity mode)


def _step(model, feats, labels_local, dev, use_got, labels_global=None, sync=True):
    from madeleine_amd import InfoNCE
    from madeleine_amd import distributed as DP
    from madeleine_amd import functional as MF
    mods = MODS5[:labels_local.shape[1]]
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.7)
    embs, toks = model({"feats": feats}, device=dev, train=True)
    loss, flag = DP.calculate_losses_dp(mods[1:], InfoNCE(temperature=0.01), MF.HipGotImpl if use_got else None, embs, toks,
                                        labels_local[:, 1:], args, labels_global_withoutHE=labels_global, use_local_loss=use_got)
    model.zero_grad()
    if sync or not hasattr(model, "no_sync"):
        loss.backward()
    else:
        with model.no_sync():
            loss.backward()
    return loss.detach(), flag


def _single(dev, use_got, labels=LABELS):
    model = _build_model(dev, labels.shape[1])
    loss, flag = _step(model, t((B, labels.shape[1], N, D), "dpg:feats"), labels, dev, use_got)
    assert flag
    return float(loss), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}


def _worker(rank, world, port, backend, use_got, ret, ddp=False, LABELS=LABELS):
    M = LABELS.shape[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from madeleine_amd import distributed as DP
        model = _build_model(dev, M)
        sync = None
        if ddp == "flat":   # what bench.py uses for N > 1: one flat all-reduce (mean) of the packed gradients after backward
            sync = DP.FlatGradSync(model, use_local_loss=use_got)
        elif ddp:   # the DistributedDataParallel wrapper (bench.py BENCH_DDP=1): bucketed gradient all-reduce (mean) inside backward
            model = DP.wrap_ddp(model, dev, use_local_loss=use_got)
        Bl = B // world
        sl = slice(rank * Bl, (rank + 1) * Bl)
        pending = DP.all_gather_labels_async(LABELS[sl, 1:])          # host-side label exchange (gloo group)
        loss, flag = _step(model, t((B, M, N, D), "dpg:feats")[sl], LABELS[sl], dev, use_got, labels_global=None)
        lab_g = pending.wait()
        assert torch.equal(lab_g, LABELS[:, 1:])
        if sync is not None:
            sync.all_reduce_mean()
        grads = {}
        for k, p in model.named_parameters():                          # what DDP does: mean over ranks
            if p.grad is None:
                continue
            g = p.grad.detach().clone()
            if not ddp:
                g = DP._all_reduce_sum(g) / world
            grads[k[7:] if k.startswith("module.") else k] = g.cpu()
        # loss value of the global batch: replicated global part + sum over ranks of the local parts (undo the W scaling)
        loss_nogot, _ = _step(model, t((B, M, N, D), "dpg:feats")[sl], LABELS[sl], dev, False, labels_global=lab_g, sync=False)
        local = ((loss - loss_nogot) / world).reshape(1).clone()
        local = DP._all_reduce_sum(local)
        if rank == 0:
            ret["loss"] = float(loss_nogot + local[0])
            ret["grads"] = {k: v.numpy() for k, v in grads.items()}
            ret["flag"] = bool(flag)
            ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_got", [False, True])
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_equal_global_batch(backend, use_got):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank (this box has %d)" % torch.cuda.device_count())
    ref_loss, ref_grads = _single(torch.device("cuda:0"), use_got)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), backend, use_got, ret), nprocs=2, join=True)
    assert ret["flag"] and ret["backend"] == backend
    assert abs(ret["loss"] - ref_loss) < 1e-4 * abs(ref_loss), (ret["loss"], ref_loss)
    top = max(float(g.norm()) for g in ref_grads.values())
    assert set(ret["grads"]) == set(ref_grads)
    for k, g in ref_grads.items():
        got = torch.from_numpy(ret["grads"][k])
        err = float((got - g).norm())
        assert err <= 1e-4 * float(g.norm()) + 1e-6 * top, (k, err, float(g.norm()))


@pytest.mark.parametrize("ddp", [True, "flat"])
@pytest.mark.parametrize("use_got", [False, True])
def test_two_ranks_under_ddp_equal_global_batch(use_got, ddp):
    """The same decomposition with the model wrapped in DistributedDataParallel exactly as bench.py wraps it for N > 1 (8-MB
    buckets, bucket views, unused-parameter detection when the local loss is off): DDP's own bucketed mean of the gradients,
    the packed all-gather in forward and the [S,6] all-reduce inside the backward of the GOT node while DDP's hooks are live."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ref_loss, ref_grads = _single(torch.device("cuda:0"), use_got)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), "gloo", use_got, ret, ddp), nprocs=2, join=True)
    assert ret["flag"]
    assert abs(ret["loss"] - ref_loss) < 1e-4 * abs(ref_loss), (ret["loss"], ref_loss)
    top = max(float(g.norm()) for g in ref_grads.values())
    for k, got in ret["grads"].items():
        g = ref_grads[k]
        err = float((torch.from_numpy(got) - g).norm())
        assert err <= 1e-4 * float(g.norm()) + 1e-6 * top, (k, err, float(g.norm()))
    assert len(ret["grads"]) >= len(ref_grads) - 2


@pytest.mark.parametrize("use_got", [False, True])
def test_four_ranks_degenerate_sharding_equal_global_batch(use_got):
    """Four gloo ranks sharing cuda:0 through the HIP kernels on the degenerate label matrix (an H&E-only rank, ranks without a case of
    a participating stain, k_global = 1 skipped, k_global = 2 split over two ranks -> one-case GOT problems with n = 2 tokens), gradient
    mean by FlatGradSync as bench.py runs it for N > 1 -- against the single-process global-batch HIP step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ref_loss, ref_grads = _single(torch.device("cuda:0"), use_got, LABELS_DEGENERATE)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(4, _free_port(), "gloo", use_got, ret, "flat", LABELS_DEGENERATE), nprocs=4, join=True)
    assert ret["flag"]
    assert abs(ret["loss"] - ref_loss) < 1e-4 * abs(ref_loss), (ret["loss"], ref_loss)
    top = max(float(g.norm()) for g in ref_grads.values())
    for k, g in ref_grads.items():
        got = torch.from_numpy(ret["grads"][k])
        err = float((got - g).norm())
        assert err <= 1e-4 * float(g.norm()) + 1e-6 * top, (k, err, float(g.norm()))


def _worker_rccl_w1(rank, port, use_got, ret, flat=False):
    """World size 1 on the RCCL backend, entered the way torch.distributed.run enters bench.py (init_from_env reading RANK /
    WORLD_SIZE / MASTER_*): every collective of the step runs as a real RCCL call on device tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("MADELEINE_DIST_BACKEND", None)
    from madeleine_amd import distributed as DP
    r, w, lr = DP.init_from_env()
    try:
        assert (r, w, lr) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl" and DP.collectives_on()
        dev = torch.device("cuda", 0)
        hg = DP.host_group()                                            # nccl default group + gloo side group
        assert hg is not None and dist.get_backend(hg) == "gloo"
        # the raw collectives on device tensors
        x = torch.arange(12, device=dev, dtype=torch.float32).view(3, 4)
        assert torch.equal(DP._all_gather_cat(x), x) and not DP._host_staged(x, None)
        assert torch.equal(DP._all_reduce_sum(x.clone()), x)
        y = x.clone().requires_grad_()
        g = DP.all_gather_replicated(y)
        assert g.grad_fn is not None and type(g.grad_fn).__name__.startswith("_AllGatherReplicatedLoss")
        g.sum().backward()
        assert torch.equal(y.grad, torch.ones_like(y))
        sync = None
        if flat:
            model = _build_model(dev)
            sync = DP.FlatGradSync(model, use_local_loss=use_got)
        else:
            model = DP.wrap_ddp(_build_model(dev), dev, use_local_loss=use_got)
        pending = DP.all_gather_labels_async(LABELS[:, 1:], hg)
        assert isinstance(pending, DP._PendingLabels)
        loss, flag = _step(model, t((B, M, N, D), "dpg:feats"), LABELS, dev, use_got, labels_global=pending.wait())
        # ... and with the labels gathered inside calculate_losses_dp (its own host exchange)
        loss2, _ = _step(model, t((B, M, N, D), "dpg:feats"), LABELS, dev, use_got, labels_global=None)
        if sync is not None:
            sync.all_reduce_mean()      # ReduceOp.AVG on the flat 20-MB buffer, RCCL
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
        dist.barrier()
        torch.cuda.synchronize()
        ret["loss"], ret["loss2"], ret["flag"] = float(loss), float(loss2), bool(flag)
        ret["grads"] = {(k[7:] if k.startswith("module.") else k): p.grad.detach().cpu().numpy()
                        for k, p in model.named_parameters() if p.grad is not None}
        ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flat", [False, True])
@pytest.mark.parametrize("use_got", [False, True])
def test_rccl_world_size_one_equals_no_process_group(use_got, flat):
    """the RCCL transport at world size 1 -- init_from_env (nccl), host_group() (gloo side group), the
    async label exchange, the packed all_gather_into_tensor on device tensors, the [S,6] all-reduce inside the GOT backward and
    DDP's bucketed gradient all-reduce (8-MB buckets, bucket views, token_projector excluded when the local loss is off) or, with
    `flat`, FlatGradSync's single all-reduce of the packed gradients (what bench.py runs for N > 1) -- must reproduce the run
    without a process group: same kernels, same order, every collective an identity."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ref_loss, ref_grads = _single(torch.device("cuda:0"), use_got)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rccl_w1, args=(_free_port(), use_got, ret, flat), nprocs=1, join=True)
    assert ret["flag"] and ret["backend"] == "nccl"
    assert abs(ret["loss"] - ref_loss) <= 1e-6 * abs(ref_loss), (ret["loss"], ref_loss)
    assert ret["loss2"] == ret["loss"]
    assert set(ret["grads"]) == set(ref_grads)
    for k, g in ref_grads.items():
        got = torch.from_numpy(ret["grads"][k])
        err = float((got - g).norm())
        assert err <= 1e-6 * float(g.norm()) + 1e-30, (k, err, float(g.norm()))
