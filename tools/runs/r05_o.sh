#!/bin/bash
# bf16 gate forward: persistent workgroup over the 4 column tiles (next tile's first chunk requested before the epilogue) vs one workgroup per tile
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05o}; mkdir -p $OUT
cd $R
for V in nopersist persist nopersist persist; do
  if [ $V = persist ]; then unset MADELEINE_BF16_GATE_NO_PERSIST; else export MADELEINE_BF16_GATE_NO_PERSIST=1; fi
  echo "== $V"; timeout 200 python tools/exp_gate_bf16.py 2>&1 | tail -6
done | tee $OUT/gate_bf16_persist.txt
unset MADELEINE_BF16_GATE_NO_PERSIST
timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
for V in nopersist persist; do
  if [ $V = persist ]; then unset MADELEINE_BF16_GATE_NO_PERSIST; else export MADELEINE_BF16_GATE_NO_PERSIST=1; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg --precision bfloat16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$V', d['ms_per_step'], d['kernel_ms'])"
done | tee $OUT/bench_bf16.txt
