#!/bin/bash
# the random +6 ms "slow mode" of a variant: tied to re-allocating the working set after empty_cache()?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
V=plain,plain,plain,plain,plain,plain,plain,plain
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
{
echo "== empty_cache between variants"
$TR --master-port 29551 tools/exp_ddp.py --variants $V --steps 10 2>&1 | grep "ms/step\|reserved"
echo "== cache kept"
$TR --master-port 29552 tools/exp_ddp.py --variants $V --steps 10 --keep-cache 2>&1 | grep "ms/step\|reserved"
} | tee $O/slowmode.txt
