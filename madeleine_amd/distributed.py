"""Data-parallel pretrain step: one process per GPU over RCCL (torch.distributed backend "nccl" on ROCm).

The reference scales with single-process nn.DataParallel (setup_components.py:185-187): the losses are
evaluated ONCE on the gathered global batch -- InfoNCE sees every other case as a negative, GOT's
thresholds are global min/max.  The MI355X-native equivalent keeps those semantics with one process per
GPU:

  * cases shard over ranks; encoder, pooling and token projection are rank-local;
  * parameter gradients: PyTorch DDP bucketed all-reduce (mean) on RCCL -- 20 MB fp32;
  * ONE autograd-aware all-gather per step of a packed per-rank payload
        [ slide embeddings B_l*M*V*512 | presence mask B_l*M | local (min,max) of the 3 GOT cost tensors per stain 6*S ]
     forms the full negative set and the global GOT thresholds; every rank then evaluates
    the same global InfoNCE, so the backward of the gather needs NO collective: rank r keeps its own slice of the
    (replicated) gradient, scaled by W so that DDP's mean over ranks reproduces the single-process global-batch
    gradient exactly.  The only other data-path collective is the [S,6] all-reduce of the threshold gradients in the
    backward of the GOT node (the thresholds are global extrema: their gradient belongs to the rank that attains them).
  * rank-local loss terms (GOT) are multiplied by W for the same reason.
  * The presence labels are HOST data the dataloader hands over with the batch; which stains take part and how many
    tokens GOT uses (n = min(k_global, 256)) must be known on the host before the loss kernels are launched.  They
    are therefore exchanged host-side (all_gather_labels_async: a gloo CPU group, started at step begin and waited
    for after the encoder forward has been queued) -- no device synchronisation; without a host group the labels fall
    back to a device all-gather + one D2H copy.

All collectives work on gloo as well (CPU tests, world_size 2).
"""
import os
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """torchrun-style init (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment). Returns (rank, world, local_rank).
    MADELEINE_DIST_BACKEND overrides the backend (e.g. gloo, to run several ranks on ONE GPU for debugging)."""
    backend = backend or os.environ.get("MADELEINE_DIST_BACKEND") or None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK and WORLD_SIZE exported): join the group even at world size 1 -- the collectives,
    # the gloo side group and DDP then run through RCCL exactly as on N GPUs (a 1-GPU box can execute that path)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        # single-node rendezvous on the loopback address: pin gloo (the host-side label exchange) to the loopback interface as well --
        # it otherwise resolves the container's hostname, which may not resolve (an explicit GLOO_SOCKET_IFNAME of the user wins)
        if os.environ.get("MASTER_ADDR") in ("127.0.0.1", "localhost", "::1"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        if torch.cuda.is_available() and world > torch.cuda.device_count():
            # debug runs with several ranks on one GPU: the processes' launches share the compute units, so the residency bound the split
            # IPOT sweeps rely on (csrc/got_impl.inc, Xch) does not hold -- keep the one-workgroup sweeps
            os.environ.setdefault("MADELEINE_GOT_NOSPLIT", "1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(device_index: int, sysfs: str = "/sys"):
    """Restricts this process's host threads to the cores of the NUMA node its GPU hangs off (one process per GPU: the launch thread,
    the pinned-memory stager and the gloo side channel then stay next to the GPU's PCIe root instead of wandering across sockets).
    The node comes from <sysfs>/bus/pci/devices/<domain:bus:device.function>/numa_node, its cores from
    <sysfs>/devices/system/node/node<k>/cpulist; the new mask is the intersection with the current affinity mask.  Returns
    {"numa_node": k, "cpus": n} or None when the topology is not exposed (containers without sysfs, numa_node = -1, an empty
    intersection) -- nothing is changed then.  MADELEINE_NO_NUMA_PIN=1 disables it."""
    if os.environ.get("MADELEINE_NO_NUMA_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        node = int(open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")).read())
        if node < 0:
            return None
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read())
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "cpus": len(cpus)}
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def wrap_ddp(model, device, use_local_loss: bool = True, bucket_cap_mb: int = 8):
    """The model under DistributedDataParallel as the data-parallel step uses it: 20 MB of fp32 gradients in 8-MB buckets (their
    all-reduce starts while the pre_attn backward is still running; the default single 25-MB bucket would only fire after the last
    gradient), bucket views instead of a copy-back.  Without the local (GOT) loss the token_projector takes no part in the graph
    (as in the reference's global-only configuration, trainer.py:36-46): it is excluded from DDP's bucket set instead of paying
    find_unused_parameters' per-step graph walk; its .grad stays None and the optimizer skips it."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if not use_local_loss:
        ignore = [n for n, _ in model.named_parameters() if n.startswith("token_projector.")]
        DDP._set_params_and_buffers_to_ignore_for_model(model, ignore)
    else:
        DDP._set_params_and_buffers_to_ignore_for_model(model, [])
    dev = torch.device(device)
    return DDP(model, device_ids=[dev.index] if dev.type == "cuda" else None, find_unused_parameters=False,
               bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)


class FlatGradSync:
    """Gradient mean over the ranks WITHOUT the DistributedDataParallel wrapper: after backward() the parameter gradients (20 MB fp32,
    ~30 tensors) are packed into ONE flat buffer by one multi-tensor copy, all-reduced by ONE RCCL call (stream-ordered; at world size > 1
    the host then reads ONE scalar back -- the ranks' agreement that nobody missed a gradient, see all_reduce_mean) and handed to the
    optimizer as views of that buffer.  On MI355X the all-reduce of 20 MB over xGMI is ~0.3-0.5 ms against a
    24-60 ms step, so hiding it behind the backward (what DDP's buckets and per-parameter hooks are for) buys < 2 %, while the wrapper's
    per-step bookkeeping measured +5 ms on this step (tools/exp_ddp.py).  Same result as DDP's mean (reference semantics:
    nn.DataParallel's summed replica gradients of a global-batch-mean loss, setup_components.py:185-187).
        sync = FlatGradSync(model, use_local_loss);  ...;  loss.backward();  sync.all_reduce_mean();  optimizer.step()
    Every packed parameter must have taken part in the step: a missing gradient (p.grad is None) raises.  DDP and the reference leave
    such a parameter's .grad None and the optimizer skips it; a zero stand-in would make AdamW apply weight decay and moment decay to
    it, and a set of None gradients that differs between ranks would be averaged silently.  The one parameter that is
    legitimately outside the graph -- the token_projector without the local loss -- is excluded through use_local_loss=False, on every
    rank alike."""

    def __init__(self, model, use_local_loss: bool = True, group=None):
        # model: an nn.Module, or an iterable of (name, leaf tensor) pairs
        named = [(n, p) for n, p in (model.named_parameters() if hasattr(model, "named_parameters") else model) if p.requires_grad]
        if not use_local_loss:
            named = [(n, p) for n, p in named if not n.startswith("token_projector.")]
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        self.group = group
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        # one extra element behind the gradients: the number of ranks that found a packed parameter without a gradient (see all_reduce_mean)
        self._buf = torch.zeros(total + 1, dtype=p0.dtype, device=p0.device)
        self.flat = self._buf[:total]
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def all_reduce_mean(self):
        missing = [self.names[i] for i, p in enumerate(self.params) if p.grad is None]
        multi = collectives_on() and dist.get_world_size(self.group) > 1
        if missing and not multi:
            raise RuntimeError(self._missing_message(missing))
        # With several ranks the verdict is agreed on THROUGH the collective: a rank that misses a gradient still enters the all-reduce
        # (its stand-ins are the buffer's previous contents, never used) with a 1 in the flag slot, and every rank raises after it --
        # a rank-local raise in front of the collective would leave the others waiting in it until the RCCL timeout.
        src = [p.grad for p, v in zip(self.params, self.views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        dst = [v for p, v in zip(self.params, self.views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if src:
            torch._foreach_copy_(dst, src)
        if collectives_on():
            W = dist.get_world_size(self.group)
            buf = self._buf if multi else self.flat
            if multi:
                self._buf[-1] = float(W if missing else 0)        # the mean over W ranks of W*[missing] = number of ranks missing
            if _host_staged(buf, self.group):
                buf.copy_(_all_reduce_sum(buf, self.group) / W)
            elif dist.get_backend(self.group) == "nccl":
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(buf, group=self.group)
                buf.div_(W)
            if multi:
                n_missing = float(self._buf[-1].item())        # one scalar read-back per step, only at world size > 1
                if n_missing > 0.5:
                    raise RuntimeError(self._missing_message(missing) if missing else
                                       "FlatGradSync: %d other rank(s) found a packed parameter without a gradient after backward; "
                                       "this step's gradient mean is void on every rank" % round(n_missing))
        for p, v in zip(self.params, self.views):
            p.grad = v

    @staticmethod
    def _missing_message(missing):
        return ("FlatGradSync: packed parameter(s) %s have no gradient after backward -- they took no part in this step's graph; exclude "
                "them when the sync is built (use_local_loss=False drops the token_projector) instead of averaging a stand-in"
                % (", ".join(missing),))


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def collectives_on() -> bool:
    """True when a process group exists: every collective of the step is then issued, also at world size 1 (where each is an
    identity) -- one code path for 1 and N ranks, and a single GPU exercises the RCCL transport."""
    return dist.is_available() and dist.is_initialized()


def _host_staged(x: torch.Tensor, group) -> bool:
    """gloo has no all_gather for device tensors: stage through the host (debug / single-GPU multi-process runs only;
    the production backend is nccl = RCCL, which takes device tensors directly)."""
    return x.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_cat(x: torch.Tensor, group=None) -> torch.Tensor:
    """[n, ...] on every rank -> [W*n, ...] (rank-major)."""
    W = dist.get_world_size(group)
    x = x.contiguous()
    if _host_staged(x, group):
        xc = x.cpu()
        out = xc.new_empty((W * xc.shape[0],) + tuple(xc.shape[1:]))
        dist.all_gather_into_tensor(out, xc, group=group)
        return out.to(x.device)
    out = x.new_empty((W * x.shape[0],) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def _all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    if _host_staged(x, group):
        xc = x.cpu()
        dist.all_reduce(xc, group=group)
        return xc.to(x.device)
    dist.all_reduce(x, group=group)
    return x


class _AllGatherReplicatedLoss(torch.autograd.Function):
    """all_gather along dim 0 whose consumer is a loss evaluated identically on every rank.

    backward: grad_in = W * grad_out[own slice]  (== the reduce-scatter-sum of W identical gradients)."""

    @staticmethod
    def forward(ctx, x, group):
        W = dist.get_world_size(group)
        ctx.rank, ctx.W, ctx.n = dist.get_rank(group), W, x.shape[0]
        return _all_gather_cat(x, group)

    @staticmethod
    def backward(ctx, g):
        sl = g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n]
        return sl * float(ctx.W), None


def all_gather_replicated(x: torch.Tensor, group=None) -> torch.Tensor:
    if not collectives_on():
        return x
    return _AllGatherReplicatedLoss.apply(x, group)


def all_gather_labels(labels: torch.Tensor, device, group=None) -> torch.Tensor:
    """[B_l, M] presence labels of every rank -> [W*B_l, M] on the host (tiny; issued at step start)."""
    if not collectives_on():
        return labels.cpu()
    x = labels.to(dtype=torch.float32).contiguous()
    if dist.get_backend(group) != "gloo":
        x = x.to(device)
    return _all_gather_cat(x, group).cpu()


class _Ready:
    def __init__(self, value):
        self.value = value

    def wait(self):
        return self.value


class _PendingLabels:
    def __init__(self, work, out):
        self.work, self.out = work, out

    def wait(self):
        self.work.wait()
        return self.out


_HOST_GROUP = {}


def host_group():
    """A gloo (CPU) process group next to the default one, for host-resident control data (presence labels).  Returns
    None when it cannot be created (the callers then fall back to the device path)."""
    if not collectives_on():
        return None
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    if "g" not in _HOST_GROUP:
        try:
            g = dist.new_group(backend="gloo")
        except Exception:  # no usable interface for gloo: device fallback
            g = None
        # every rank must take the same path afterwards (one rank on the device all-gather while the others wait on gloo would
        # deadlock): agree on the outcome over the default group
        ok = torch.tensor([0 if g is None else 1], dtype=torch.int32,
                          device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        _HOST_GROUP["g"] = g if int(ok.item()) == 1 else None
    return _HOST_GROUP["g"]


def all_gather_labels_async(labels: torch.Tensor, hgroup=None):
    """Starts the host-side exchange of the [B_l, M] presence labels; .wait() returns the [W*B_l, M] CPU tensor.  No
    device work, no device synchronisation (issue at step begin, wait after the encoder forward has been queued)."""
    if not collectives_on():
        return _Ready(labels.detach().cpu().float())
    if hgroup is None:
        hgroup = host_group()
    if hgroup is None:
        return _Ready(all_gather_labels(labels, torch.device("cuda", torch.cuda.current_device())))
    x = labels.detach().cpu().to(torch.float32).contiguous()
    out = x.new_empty((dist.get_world_size(hgroup) * x.shape[0],) + tuple(x.shape[1:]))
    return _PendingLabels(dist.all_gather_into_tensor(out, x, group=hgroup, async_op=True), out)


def _pack_embeddings(wsi_embs, modalities):
    parts = [wsi_embs[m][..., 0] if m == "HE" else wsi_embs[m] for m in modalities]      # each [B_l,V,512]
    Bl, V, D = parts[0].shape
    return torch.stack(parts, dim=1).reshape(Bl * len(modalities) * V * D), (Bl, V, D)


def _unpack_embeddings(flat_rows, modalities, geom):
    """flat_rows [W, B_l*M*V*D] (rank-major) -> the global dict in the reference's shapes (HE expanded over M-1)."""
    Bl, V, D = geom
    M = len(modalities)
    full = flat_rows.reshape(-1, M, V, D)
    out = {}
    for i, m in enumerate(modalities):
        e = full[:, i]
        out[m] = e.unsqueeze(3).expand(-1, -1, -1, M - 1) if m == "HE" else e
    return out


def gather_packed(wsi_embs: Dict[str, torch.Tensor], modalities: Sequence[str], labels_local: torch.Tensor,
                  extrema_local: Optional[torch.Tensor], group=None):
    """THE data-path collective of a step: one autograd-aware all-gather of
        [ slide embeddings B_l*M*V*512 | presence mask B_l*M | GOT threshold extrema 6*S ]   per rank.
    Returns (global embedding dict, presence mask [W*B_l, M] on the device, extrema of every rank [W, S, 6] or None).
    Gradients flow to the embedding part only (own slice x W, no collective in backward)."""
    W = world_size(group)
    if not collectives_on():
        from .functional import h2d
        return wsi_embs, h2d(labels_local, wsi_embs["HE"].device), None if extrema_local is None else extrema_local.unsqueeze(0)
    emb, geom = _pack_embeddings(wsi_embs, modalities)
    dev, dt = emb.device, emb.dtype
    if extrema_local is not None and dt not in (torch.float32, torch.float64):
        # the thresholds travel in the embedding payload: a narrower dtype would round them away from the local cost extrema
        # the GOT backward routes their gradient by
        raise TypeError("gather_packed: slide embeddings must be float32 / float64 when GOT extrema are packed (got %s)" % dt)
    from .functional import h2d
    lab = h2d(labels_local, dev, dt).reshape(-1)
    ext = extrema_local.to(dt).reshape(-1) if extrema_local is not None else emb.new_zeros(0)
    payload = torch.cat([emb, lab, ext]).unsqueeze(0)                               # [1, P]
    full = all_gather_replicated(payload, group)                                    # [W, P]
    n_e, n_l = emb.numel(), lab.numel()
    embs_g = _unpack_embeddings(full[:, :n_e], modalities, geom)
    labels_g = full[:, n_e:n_e + n_l].detach().reshape(W * labels_local.shape[0], -1)
    ext_all = full[:, n_e + n_l:].detach().reshape(W, -1, 6) if extrema_local is not None else None
    return embs_g, labels_g, ext_all


def gather_slide_embeddings(wsi_embs: Dict[str, torch.Tensor], modalities: Sequence[str], group=None):
    """Slide embeddings only (one all-gather of the packed [B_l, M*V*512] payload); see gather_packed."""
    if not collectives_on():
        return wsi_embs
    emb, geom = _pack_embeddings(wsi_embs, modalities)
    return _unpack_embeddings(all_gather_replicated(emb.unsqueeze(0), group), modalities, geom)


# --------------------------------------------------------------------------------------------------
# local (GOT) loss with global-batch semantics
# --------------------------------------------------------------------------------------------------
_SIDE_STREAMS = {}


class _fan_out:
    """Run independent launches concurrently and join them back into the current stream:
        with _fan_out(device, n) as lanes:
            with lanes(i): launch_i()
    Lane 0 IS the current stream; lanes 1 .. n-1 are side HIP streams that first wait for the work already queued on the current
    stream; on exit the current stream waits for every side lane.  n lanes = n streams in total: with the four stains of the reference's
    datasets that is 4 streams on ROCm's default 4 hardware queues (GPU_MAX_HW_QUEUES) -- one queue per chain.  (Round 3 used n side
    streams + the main one and raised GPU_MAX_HW_QUEUES to 8; with 8 queues a host that runs ahead of the device pays for it: every
    small kernel behind pending queues takes 3-4x as long -- +8 ms per config-3 step, +4 ms under DDP; profiles/r04_hw_queues_*.txt.)
    Tensors allocated inside a lane are only consumed after the join (same allocator stream semantics as torch.cuda.stream +
    wait_stream).  On CPU tensors (gloo tests) it is a no-op."""

    def __init__(self, device, n):
        self.on = torch.device(device).type == "cuda" and n > 1
        self.device, self.n = device, n

    def __enter__(self):
        if self.on:
            pool = _SIDE_STREAMS.setdefault(str(self.device), [])   # ONE growing pool per device: never more side streams than lanes
            while len(pool) < self.n - 1:
                pool.append(torch.cuda.Stream(device=self.device))
            self.streams = pool[:self.n - 1]
            self.main = torch.cuda.current_stream(self.device)
            self.used = set()
            # the side lanes depend on what the current stream held BEFORE the fan-out, not on lane 0's own launches
            self.start = torch.cuda.Event()
            self.start.record(self.main)
            # The split IPOT sweeps (two workgroups per case that wait for each other, csrc/got_impl.inc) size every launch so that all of
            # its workgroups can be resident together; several chains launched side by side on different streams void that bound: keep the
            # one-workgroup sweeps for launches issued inside a fan-out (the launcher reads the variable at every launch).
            self.nosplit_was = os.environ.get("MADELEINE_GOT_NOSPLIT")
            os.environ["MADELEINE_GOT_NOSPLIT"] = "1"
        return self._lane

    def _lane(self, i):
        import contextlib
        if not self.on or i == 0:
            return contextlib.nullcontext()
        st = self.streams[i - 1]
        st.wait_event(self.start)
        self.used.add(i - 1)
        return torch.cuda.stream(st)

    def __exit__(self, *exc):
        if self.on:
            if self.nosplit_was is None:
                os.environ.pop("MADELEINE_GOT_NOSPLIT", None)
            else:
                os.environ["MADELEINE_GOT_NOSPLIT"] = self.nosplit_was
            for i in self.used:
                self.main.wait_stream(self.streams[i])
        return False


def _reduce_extrema(ext_all: torch.Tensor) -> torch.Tensor:
    """[W, S, 6] (min,max | min,max | min,max per problem) of every rank -> the global [S, 6]."""
    even = (torch.arange(6, device=ext_all.device) % 2 == 0)
    return torch.where(even, ext_all.amin(dim=0), ext_all.amax(dim=0))


def got_local_extrema(problems, impl=None) -> torch.Tensor:
    """[S, 6] threshold extrema of this rank's share of every GOT problem (+-inf for a rank that owns no case of it)."""
    if impl is None:
        from .functional import HipGotImpl as impl  # noqa: N813
    inf = float("inf")
    dev, dt = problems[0][0].device, problems[0][0].dtype
    # (on the caller's stream: fanning these small cost-matrix passes out over the per-stain streams saved 0.2 ms in the four-stain lab
    # and cost 6.5 ms per config-3 step -- their workspaces then come from the side streams' allocator pools)
    live = [(V.contiguous(), Q.contiguous()) for V, Q in problems if V.shape[0] > 0]
    if len(live) == len(problems) and getattr(impl, "can_batch", None) and impl.can_batch(live):
        return impl.extrema_multi(live)   # two launches for all stains
    empty = None
    out = []
    for V, Q in problems:
        if V.shape[0] > 0:
            out.append(impl.extrema(V.contiguous(), Q.contiguous()))
        else:
            if empty is None:   # (min, max) x 3 = (+inf, -inf) x 3, built on the device: a host constant would be a blocking copy
                empty = torch.stack([torch.full((3,), inf, device=dev, dtype=dt), torch.full((3,), -inf, device=dev, dtype=dt)], dim=1).reshape(6)
            out.append(empty)
    return torch.stack(out)


GOT_LANES = int(os.environ.get("MADELEINE_GOT_LANES", "4"))   # HIP streams (the caller's included) the GOT chains of a step are spread over


def _lane_of(shapes):
    """Lane per problem for _fan_out: the chains (cost ~ k n^3) are packed onto min(#problems, GOT_LANES) lanes, heaviest first onto the
    lightest lane; the LIGHTEST lane is lane 0 = the caller's stream, whose later launches (the InfoNCE section, the loss sum) then
    wait for the shortest queue.  ROCm gives a process 4 hardware queues by default: every stream beyond that -- a prefetcher's, RCCL's
    -- makes two lanes share a queue and their chains run one after the other (profiles/r04i_c4_streams_vs_lanes.txt), so fewer lanes
    can be the faster choice in such a process (MADELEINE_GOT_LANES)."""
    live = [s for s, (vs, _qs) in enumerate(shapes) if vs[0] > 0]
    lane = [0] * len(shapes)
    if not live:
        return lane
    cost = {s: float(shapes[s][0][0]) * float(shapes[s][0][1]) ** 3 for s in live}
    n_l = max(1, min(len(live), GOT_LANES))
    load, members = [0.0] * n_l, [[] for _ in range(n_l)]
    for s in sorted(live, key=lambda q: -cost[q]):
        i = min(range(n_l), key=lambda j: load[j])
        load[i] += cost[s]
        members[i].append(s)
    order = sorted(range(n_l), key=lambda j: load[j])          # lightest bin -> lane 0
    for new_lane, j in enumerate(order):
        for s in members[j]:
            lane[s] = new_lane
    return lane


def _rows_of(t, live, S):
    """t[live] without a host-side index tensor (live: ascending Python ints)."""
    return t if len(live) == S else torch.stack([t[i] for i in live])


def _scatter_rows(part, live, S):
    """[S, ...] with part's rows at `live` and zeros elsewhere, again without an index tensor."""
    if len(live) == S:
        return part
    zero = torch.zeros_like(part[0])
    pos = {s: i for i, s in enumerate(live)}
    return torch.stack([part[pos[s]] if s in pos else zero for s in range(S)])


class _GOTMulti(torch.autograd.Function):
    """S GOT problems (one per stain) in ONE autograd node with global-batch thresholds `ext` [S,6].

    forward : forwards of every problem with the given thresholds (side streams)
    backward: reverse sweeps of every problem -> one all-reduce of the threshold gradients [S,6] -> finish
    Ranks that own no case of a stain pass an empty (k = 0) problem: they still take part in the collective.
    With world size 1 and ext = the batch's own extrema this is S independent reference-semantics GOT calls."""

    @staticmethod
    def forward(ctx, impl, group, ext, *tensors):
        S = len(tensors) // 2
        probs = [(tensors[2 * s].contiguous(), tensors[2 * s + 1].contiguous()) for s in range(S)]
        dev = tensors[0].device
        outs, states = [], []
        ctx.shapes = [(V.shape, Q.shape) for V, Q in probs]
        live = [s for s in range(S) if probs[s][0].shape[0] > 0]
        ctx.batched = None
        if getattr(impl, "can_batch", None) and impl.can_batch([probs[s] for s in live]):
            # ONE launch sequence for all stains on the caller's stream (mdl_got_*_multi): no side streams, no hardware queues to share
            # (rows are picked / scattered by stacking views: indexing a device tensor with a Python list is a pageable H2D copy of the
            # index, i.e. a host synchronisation in the middle of the loss section -- 1.2 ms of idle device per config-3 step)
            o, st = impl.forward_multi([probs[s] for s in live], _rows_of(ext, live, S))
            ctx.impl, ctx.group, ctx.batched, ctx.live = impl, group, st, live
            return _scatter_rows(o, live, S)
        ctx.lane = lane = _lane_of(ctx.shapes)
        ctx.n_lanes = max(lane) + 1
        with _fan_out(dev, ctx.n_lanes) as lanes:   # stains are independent: one HIP stream each (k <= 32 workgroups per problem)
            for s, (V, Q) in enumerate(probs):
                if V.shape[0] == 0:
                    outs.append(torch.zeros(2, device=dev, dtype=tensors[0].dtype))
                    states.append(None)
                else:
                    with lanes(lane[s]):
                        o, st = impl.forward(V, Q, ext[s])
                    outs.append(o)
                    states.append(st)
        ctx.impl, ctx.group, ctx.states = impl, group, states
        return torch.stack(outs)

    @staticmethod
    def backward(ctx, d_outs):
        impl = ctx.impl
        dev = d_outs.device
        d_outs = d_outs.contiguous()
        if ctx.batched is not None:
            live, S = ctx.live, len(ctx.shapes)
            dmm = _scatter_rows(impl.backward_begin_multi(ctx.batched, _rows_of(d_outs, live, S)), live, S)
            if collectives_on() and ctx.group is not _LOCAL:
                dmm = _all_reduce_sum(dmm, ctx.group)
            pairs = impl.backward_finish_multi(ctx.batched, _rows_of(dmm, live, S))
            grads = []
            for s in range(S):
                if s in live:
                    grads += list(pairs[live.index(s)])
                else:
                    grads += [d_outs.new_zeros(ctx.shapes[s][0]), d_outs.new_zeros(ctx.shapes[s][1])]
            return (None, None, None) + tuple(grads)
        states = ctx.states
        parts = []
        with _fan_out(dev, ctx.n_lanes) as lanes:
            for s, st in enumerate(states):
                if st is None:
                    parts.append(torch.zeros(6, device=dev, dtype=d_outs.dtype))
                else:
                    with lanes(ctx.lane[s]):
                        parts.append(impl.backward_begin(st, d_outs[s]))
        dmm = torch.stack(parts)
        if collectives_on() and ctx.group is not _LOCAL:
            dmm = _all_reduce_sum(dmm, ctx.group)
        grads = []
        with _fan_out(dev, ctx.n_lanes) as lanes:
            for s, st in enumerate(states):
                if st is None:
                    grads += [d_outs.new_zeros(ctx.shapes[s][0]), d_outs.new_zeros(ctx.shapes[s][1])]
                else:
                    with lanes(ctx.lane[s]):
                        grads += list(impl.backward_finish(st, dmm[s]))
        return (None, None, None) + tuple(grads)


_LOCAL = object()   # group sentinel: this process's problems only, no collective even when a process group exists


def got_multi(problems, impl=None, group=None, extrema=None, local=False) -> torch.Tensor:
    """problems: list of (V, Q) token tensors [k_s, n_s, d] (already sub-sampled) -> [S, 2] = (WD sum, GWD sum).
    `extrema` [S,6]: the global-batch thresholds (from gather_packed); None -> computed here (one [S,6] all-gather).
    local=True: S independent reference-semantics GOT calls of THIS process (thresholds from its own problems, no collective in either
    direction) -- trainer.calculate_losses' per-stain GOT terms, run as concurrent chains."""
    if impl is None:
        from .functional import HipGotImpl as impl  # noqa: N813
    if local:
        group = _LOCAL
    if extrema is None:
        extrema = got_local_extrema(problems, impl)
        if collectives_on() and not local:
            extrema = _reduce_extrema(_all_gather_cat(extrema, group).view(world_size(group), -1, 6))
    flat = []
    for V, Q in problems:
        flat += [V, Q]
    return _GOTMulti.apply(impl, group, extrema, *flat)


_STEP = [0]


def _shared_he_tokens(he_all):
    """MADELEINE returns the H&E tokens as an expand() over the stain axis (Model.py:153-155 repeats them): every slice
    [:, :, :, s_idx] is the same memory.  ONE select shared by all stains -- otherwise each stain's backward materialises a zero
    [B,N,128,M-1] tensor and the expand's backward sums them (dozens of dense fills / adds per step at 5 stains).  None when the
    tensor is not such a view."""
    if he_all.dim() == 4 and he_all.stride(3) == 0:
        return he_all.select(3, 0)
    return None


def calculate_losses_dp(STAINS, loss_fn_interMod, got_impl, wsi_embs, token_embs, modality_labels_withoutHE, args,
                        labels_global_withoutHE=None, group=None, subsample=256, shared_seed=0, use_local_loss=True,
                        loss_fn_intraMod=None):
    """Data-parallel counterpart of calculate_losses (trainer.py:20-77) with the reference's global-batch semantics.

    wsi_embs / token_embs / modality_labels_withoutHE are this rank's shard; labels_global_withoutHE the [W*B_l, M-1]
    presence labels of the global batch on the HOST (all_gather_labels_async(...).wait(); gathered here when None).
    Returns (loss, flag) where `loss` is the tensor to call .backward() on under DDP (gradient mean over ranks): the
    replicated global InfoNCE (+ intra-modality terms) plus W x (this rank's GOT sum); averaged over ranks its gradient
    equals the single-process global-batch gradient of  sum_stains [InfoNCE + w * GOT]  (SURVEY.md section 8(e)).
    One data-path collective in forward (gather_packed) and one [S,6] all-reduce in backward."""
    from .functional import h2d
    from .trainer import calculate_losses
    W = world_size(group)
    dev = wsi_embs["HE"].device
    labels_l = modality_labels_withoutHE.detach().cpu()
    if labels_global_withoutHE is not None:
        labels_g = labels_global_withoutHE
    elif not collectives_on():
        labels_g = labels_l
    else:
        if group is not None and group is not dist.group.WORLD:
            # the host label exchange runs over the world (gloo) group: with a sub-group the caller must supply the labels of
            # exactly the ranks the embeddings are gathered over
            raise ValueError("calculate_losses_dp(group=<sub-group>) needs labels_global_withoutHE from the caller")
        labels_g = all_gather_labels_async(labels_l).wait()
    mods = ["HE"] + list(STAINS)

    # the local (GOT) problems of this step: fixed by the HOST copy of the global labels, identical on every rank
    problems = []
    he_shared, first_local = None, -1
    if use_local_loss and got_impl is not None and any(int(labels_g[:, s].bool().sum()) > 1 for s in range(len(STAINS))):
        _STEP[0] += 1
        for s_idx, stain in enumerate(STAINS):
            k_g = int(labels_g[:, s_idx].bool().sum())
            if k_g <= 1:
                continue
            # reference: randperm(k_global)[:subsample] (loss.py:282) = the first min(k_g, subsample) tokens, any order
            if k_g <= subsample:
                tok_idx = torch.arange(k_g)
            else:
                gen = torch.Generator().manual_seed(int(shared_seed) * 1000003 + _STEP[0] * 131 + s_idx)
                tok_idx = torch.randperm(k_g, generator=gen)[:subsample]
            if int(tok_idx.max()) >= token_embs["HE"].shape[1]:
                raise ValueError("GOT sub-samples token indices randperm(k)[:%d] with k = %d participating cases, but the bags "
                                 "carry only %d tokens (reference quirk, loss.py:282)" % (subsample, k_g, token_embs["HE"].shape[1]))
            rows = h2d(labels_l[:, s_idx].bool().nonzero(as_tuple=True)[0], dev)
            if first_local < 0:
                first_local, he_shared = s_idx, _shared_he_tokens(token_embs["HE"])
            he = he_shared if he_shared is not None else token_embs["HE"][:, :, :, s_idx]
            st = token_embs[stain]
            # token axis first (a view when the indices are the first k_g tokens -- the common case), then the participating cases:
            # the gathers and their backward then touch [B, n, 128] instead of [B, N, 128]
            if k_g <= subsample:
                he, st = he[:, :k_g], st[:, :k_g]
            else:
                tok_idx = h2d(tok_idx, dev)
                he, st = he.index_select(1, tok_idx), st.index_select(1, tok_idx)
            he, st = he.index_select(0, rows), st.index_select(0, rows)
            problems.append((he if he.dtype == torch.float64 else he.float(), st if st.dtype == torch.float64 else st.float()))
    ext_local = got_local_extrema(problems, got_impl) if problems else None

    # THE collective: slide embeddings + presence mask + GOT extrema in one payload
    pad = torch.ones(labels_l.shape[0], 1, dtype=labels_l.dtype)                      # H&E column: always present
    embs_g, _mask_dev, ext_all = gather_packed(wsi_embs, mods, torch.cat([pad, labels_l], dim=1), ext_local, group)
    # The GOT chains are queued FIRST (lane 0 on this stream, the other stains on side streams): the ~60 small launches of the InfoNCE
    # section then run behind lane 0's chain, inside the time the longer lanes need anyway, instead of in front of all four
    # (profiles/r04e_c4_got_concurrency.txt: 0.7 ms of a config-4 rank step).  Same values: nothing below depends on the order.
    outs = got_multi(problems, got_impl, group, extrema=_reduce_extrema(ext_all)) if problems else None     # [S,2]
    loss_g, flag = calculate_losses(STAINS, loss_fn_interMod, None, loss_fn_intraMod, embs_g, None, labels_g, args)
    if not flag or outs is None:
        return loss_g, flag
    local = (outs[:, 1] + outs[:, 0]).sum() * args.local_loss_weight
    loss = (loss_g if torch.is_tensor(loss_g) else 0.0) + float(W) * local
    return loss, flag
