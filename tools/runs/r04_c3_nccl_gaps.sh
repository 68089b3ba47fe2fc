#!/bin/bash
# device idle gaps of a config-3 step under torchrun (RCCL, world size 1) against the plain step: where do the collectives cost time?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_txt; mkdir -p $O
X="--config c3 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg"
{
for v in nccl plain; do
  rm -rf /tmp/prof_c3n
  if [ $v = nccl ]; then
    BENCH_NO_TIMER=1 rocprofv3 --kernel-trace -d /tmp/prof_c3n -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 1 $X > /tmp/prof_c3n.log 2>&1
  else
    BENCH_NO_TIMER=1 rocprofv3 --kernel-trace -d /tmp/prof_c3n -- python $R/bench.py $X > /tmp/prof_c3n.log 2>&1
  fi
  for db in $(ls -S /tmp/prof_c3n/*/*.db | head -1); do
    echo "== $v"; python $R/tools/rocpd_gaps.py $db FusedAdam 6 8 25
  done
done
} > $O/${1:-r04m}_c3_nccl_gaps.txt 2>&1
cat $O/${1:-r04m}_c3_nccl_gaps.txt | cut -c1-170
