#!/bin/bash
# host glue of the loss section (identity gathers / padding skipped) A/B on one box: c2 and c3 all-present steps
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05t}; mkdir -p $OUT
cd $R
for C in c2 c3; do
for V in always skip always skip; do
  if [ $V = skip ]; then unset MADELEINE_LOSS_GATHER_ALWAYS; else export MADELEINE_LOSS_GATHER_ALWAYS=1; fi
  timeout 300 python bench.py --config $C --steps 12 --warmup 4 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$C $V', d['ms_per_step'])"
done; done | tee $OUT/bench.txt
