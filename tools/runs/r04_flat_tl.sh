#!/bin/bash
# device idle gaps per step: flat gradient sync against no sync (why does a 0.03-ms collective cost milliseconds inside the step?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
for v in flat flat_nocomm; do
rm -rf /tmp/prof_fl
rocprofv3 --kernel-trace -d /tmp/prof_fl -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 $R/tools/exp_ddp.py --variants $v --steps 3 > /tmp/prof_fl.log 2>&1
grep "ms/step" /tmp/prof_fl.log
for db in /tmp/prof_fl/*/*.db; do
  echo "== $v (marker = the fused AdamW lerp kernel, one per step)"
  python $R/tools/rocpd_gaps.py $db LerpFunctor 4 5 30
  python $R/tools/rocpd_gaps.py $db LerpFunctor 5 6 30
done
done > $O/flat_gaps.txt 2>&1
cat $O/flat_gaps.txt
