#!/bin/bash
# bench.py through its N = 8 path on ONE GPU box: eight gloo ranks share cuda:0 (world-size-8 logic: host label exchange, packed
# all-gather, 256-case global InfoNCE, GOT with n = min(k_global, 256), DDP over 8 ranks) -- a functional check, not a timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in c2; do   # (c3 needs ~40 GB per rank: eight ranks do not fit one 288 GB device)
MADELEINE_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 8 --steps 3 --warmup 1 --config $cfg --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg > gpurun_out/r03_n8_gloo_$cfg.json 2> gpurun_out/r03_n8_gloo_$cfg.err
echo "rc=$?"; tail -c 700 gpurun_out/r03_n8_gloo_$cfg.json | cut -c 1-600; tail -3 gpurun_out/r03_n8_gloo_$cfg.err | cut -c 1-300
done
