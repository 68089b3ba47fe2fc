#!/bin/bash
# bench.py through its N = 2 path (one GPU box: two gloo ranks share cuda:0) for c2 and c3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in c2 c3; do
MADELEINE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --config $cfg --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg > gpurun_out/r03_n2_gloo_$cfg.json 2> gpurun_out/r03_n2_gloo_$cfg.err
echo "rc=$?"; tail -c 600 gpurun_out/r03_n2_gloo_$cfg.json | cut -c 1-400; tail -3 gpurun_out/r03_n2_gloo_$cfg.err
done
