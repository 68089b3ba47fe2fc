#!/bin/bash
# bf16 mode after the Linear rule change: step x3 (round-4 rule vs new), bf16 + kernel tests
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05w}; mkdir -p $OUT
cd $R
for V in 1024 512 1024 512 1024 512; do
  export MADELEINE_BF16_LIN256_MINK=$V
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg --precision bfloat16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms']; print('mink $V', d['ms_per_step'], {n: k[n] for n in ('linear_fwd','linear_bwd','gate_fwd')})"
done | tee $OUT/bench.txt
unset MADELEINE_BF16_LIN256_MINK
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_hip_kernels.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
