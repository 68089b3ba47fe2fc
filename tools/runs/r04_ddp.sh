#!/bin/bash
# DDP / RCCL overhead at world size 1 on one box: the same c2 step without a process group, under torch.distributed.run (nccl + DDP),
# and the kernel-trace summary of the latter (which RCCL kernels run, how long, what they do to the step).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04b; mkdir -p $O
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs --steps 20 --warmup 5"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
short() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('collective_backend'))" $1; }
BENCH_NO_TIMER=1 python bench.py $B > $O/plain_notimer.json 2>/dev/null; short $O/plain_notimer.json
BENCH_NO_TIMER=1 $TR --master-port 29512 bench.py --gpus 1 $B > $O/w1_nccl_notimer.json 2>/dev/null; short $O/w1_nccl_notimer.json
BENCH_NO_TIMER=1 MADELEINE_DIST_BACKEND=gloo $TR --master-port 29513 bench.py --gpus 1 $B > $O/w1_gloo_notimer.json 2>/dev/null; short $O/w1_gloo_notimer.json
python bench.py $B > $O/plain_timer.json 2>/dev/null; short $O/plain_timer.json
BENCH_NO_TIMER=1 python bench.py --config c3 $B > $O/c3_plain_notimer.json 2>/dev/null; short $O/c3_plain_notimer.json
BENCH_NO_TIMER=1 $TR --master-port 29514 bench.py --gpus 1 --config c3 $B > $O/c3_w1_nccl_notimer.json 2>/dev/null; short $O/c3_w1_nccl_notimer.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_w1
BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_w1 -- $TR --master-port 29515 $R/bench.py --gpus 1 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs --steps 3 --warmup 1 > /tmp/prof_w1.log 2>&1
for db in /tmp/prof_w1/*/*.db; do python $R/tools/rocpd_summary.py $db 45; done > $R/$O/w1_nccl_kernel_stats.txt 2>&1
tail -3 /tmp/prof_w1.log
