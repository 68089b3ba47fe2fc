#!/bin/bash
# bf16 TN products on the 256 x 256 tile (tn256_mainloop): MADELEINE_BF16_TN256=0|1 per-product times, bf16 tests, c2 bf16 step
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05tn}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do for V in 0 1; do
  export MADELEINE_BF16_TN256=$V
  echo "== TN256=$V"; true
done; done | tee $OUT/linear.txt
for V in 0 1 0 1; do
  export MADELEINE_BF16_TN256=$V
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg --precision bfloat16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms']; print('TN256=$V', d['ms_per_step'], {n: k[n] for n in ('linear_bwd','gate_bwd_gemm')}, d['config'].get('final_loss'))"
done | tee $OUT/bench.txt
