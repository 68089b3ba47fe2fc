#!/bin/bash
# bench.py through its N > 1 path on a one-GPU box: gloo ranks sharing cuda:0 (N = 2: c2 and c3; N = 8: c2), and N = 1 under torchrun on RCCL
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04n; mkdir -p $O
X="--no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg"
for cfg in c2 c3; do
MADELEINE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --config $cfg $X > $O/n2_gloo_shared_gpu_$cfg.json 2> $O/n2_$cfg.err
echo "rc=$?"; python -c "import json,sys; d=json.loads(open('$O/n2_gloo_shared_gpu_$cfg.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'])"; tail -2 $O/n2_$cfg.err
done
MADELEINE_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --steps 3 --warmup 1 --config c2 $X > $O/n8_gloo_shared_gpu_c2.json 2> $O/n8_c2.err
echo "rc=$?"; python -c "import json,sys; d=json.loads(open('$O/n8_gloo_shared_gpu_c2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'])"; tail -2 $O/n8_c2.err
for cfg in c2 c3; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 3 --config $cfg $X > $O/n1_torchrun_nccl_$cfg.json 2> $O/n1_$cfg.err
echo "rc=$?"; python -c "import json,sys; d=json.loads(open('$O/n1_torchrun_nccl_$cfg.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'])"
python bench.py --steps 20 --warmup 3 --config $cfg $X 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', d['value'], d['ms_per_step'])"
done
