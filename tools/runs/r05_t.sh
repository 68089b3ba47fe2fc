#!/bin/bash
# host glue of the loss section A/B on one box (MADELEINE_LOSS_GATHER_ALWAYS=1 = round-4 behaviour): c3, c3 all-present, c2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05t}; mkdir -p $OUT
cd $R
for C in c3 c2; do
for V in always skip always skip; do
  if [ $V = skip ]; then unset MADELEINE_LOSS_GATHER_ALWAYS; else export MADELEINE_LOSS_GATHER_ALWAYS=1; fi
  timeout 300 python bench.py --config $C --steps 12 --warmup 4 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$C $V', d['ms_per_step'], d['config'].get('final_loss'))"
done; done | tee $OUT/bench.txt
unset MADELEINE_LOSS_GATHER_ALWAYS
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
