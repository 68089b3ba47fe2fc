#!/bin/bash
# does the in-step collective's cost depend on the number of hardware queues the HIP streams map to?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
for Q in 1 2 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$Q"
  GPU_MAX_HW_QUEUES=$Q python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$Q tools/exp_ddp.py --variants plain,flat,flat_nocomm,ddp,flat --steps 12 2>&1 | grep "^{"
done | tee $O/hwq.txt
