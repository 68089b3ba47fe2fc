#!/usr/bin/env python3
"""Where does the data-parallel wrapper's time go?  The config-2 step at world size 1 on the RCCL backend (launched by
torch.distributed.run) under several gradient-synchronisation variants, same process, same box:
    plain      no gradient collective at all (the collectives of the loss still run)
    ddp        distributed.wrap_ddp as bench.py uses it
    ddp_*      DDP with other settings
    flat       no DDP: parameter gradients packed into ONE flat buffer after backward, one all-reduce, unpacked (distributed.FlatGradSync)
Usage: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/exp_ddp.py [--config c2|c3]"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BN  # noqa: E402
from madeleine_amd import InfoNCE, MADELEINE  # noqa: E402
from madeleine_amd import distributed as D  # noqa: E402
from madeleine_amd import functional as MF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--keep-cache", action="store_true", help="do not empty the caching allocator between variants")
    ap.add_argument("--variants", default="plain,ddp,flat,flat_nocomm,plain")
    a = ap.parse_args()
    rank, world, local_rank = D.init_from_env()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    B, M, N, Dm, use_got, stain_enc = BN.CONFIGS[a.config]
    mods = BN.MODS5[:M]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    feats = torch.randn(B, M, N, Dm, device=dev, generator=gen)
    labels = torch.ones(B, M)
    if M > 2:
        rates = torch.tensor([1.0, 0.46, 0.73, 0.73, 0.73][:M])
        labels = (torch.rand(B, M, generator=torch.Generator().manual_seed(77 + rank)) < rates).float()
        labels[:, 0] = 1
        feats = feats * labels.to(dev)[:, :, None, None]
    data = {"feats": feats, "modality_labels": labels}
    crit = InfoNCE(temperature=0.001)
    got_impl = MF.HipGotImpl if use_got else None
    largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
    hgroup = D.host_group()
    from torch.nn.parallel import DistributedDataParallel as DDP
    out = {}
    for v in a.variants.split(","):
        torch.manual_seed(42)
        model = MADELEINE(BN.make_cfg(M, Dm), stain_encoding=stain_enc).to(dev).train()
        net, sync = model, None
        ign = [] if use_got else [n for n, _ in model.named_parameters() if n.startswith("token_projector.")]
        if v == "ddp":
            net = D.wrap_ddp(model, dev, use_local_loss=use_got)
        elif v.startswith("ddp_"):
            DDP._set_params_and_buffers_to_ignore_for_model(model, ign)
            net = DDP(model, device_ids=[dev.index], broadcast_buffers="nobcast" not in v,
                      bucket_cap_mb=(64 if "1bucket" in v else 8), gradient_as_bucket_view=True, static_graph="static" in v)
        elif v.startswith("flat"):
            sync = D.FlatGradSync(model, use_local_loss=use_got)
            if v == "flat_nocomm":      # pack + view hand-over only
                sync.all_reduce_mean.__func__  # noqa: B018
                import types
                def _no(self=sync):
                    on = D.collectives_on
                    D.collectives_on = lambda: False
                    try:
                        D.FlatGradSync.all_reduce_mean(self)
                    finally:
                        D.collectives_on = on
                sync.all_reduce_mean = _no
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4)

        def step():
            opt.zero_grad(set_to_none=sync is None)
            pending = D.all_gather_labels_async(labels[:, 1:], hgroup)
            embs, toks = net(data, device=dev)
            loss, _ = D.calculate_losses_dp(mods[1:], crit, got_impl, embs, toks, labels[:, 1:], largs,
                                            labels_global_withoutHE=pending.wait(), use_local_loss=use_got)
            loss.backward()
            if sync is not None:
                sync.all_reduce_mean()
            opt.step()
            return loss
        for _ in range(4):
            step()
        torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        torch.distributed.barrier()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        out[v + ("" if v not in out else "_again")] = round(ms, 3)
        print(v, round(ms, 3), "ms/step  loss", float(loss.detach()), flush=True)
        del net, model, opt
        if not a.keep_cache:
            torch.cuda.empty_cache()
        print("   reserved GiB", round(torch.cuda.memory_reserved() / 2 ** 30, 2), "device mallocs", torch.cuda.memory_stats().get("num_device_alloc", 0), flush=True)
    # the collective alone: 20 MB all-reduce (AVG) on the RCCL stream, back to back
    flat = torch.zeros(5_000_000, device=dev)
    for _ in range(3):
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.AVG)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20):
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.AVG)
    e1.record()
    host_ms = 1e3 * (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    out["allreduce_20MB_device_ms"] = round(e0.elapsed_time(e1) / 20, 4)
    out["allreduce_20MB_host_enqueue_ms"] = round(host_ms, 4)
    print(out)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
