#!/bin/bash
# gate forward persistence modes (0 = one tile per workgroup, 1 = 4 column tiles, 2 = 4 token tiles), split and bf16 kernels, same box
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05p}; mkdir -p $OUT
cd $R
for V in 0 1 2 0 1 2; do
  export MADELEINE_GATE_PERSIST=$V
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('split pmode $V', d['ms_per_step'], 'gate_fwd', d['kernel_ms']['gate_fwd'])"
done | tee $OUT/bench_split.txt
unset MADELEINE_GATE_PERSIST
for V in 0 1 2 0 1 2; do
  export MADELEINE_BF16_GATE_PERSIST=$V
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg --precision bfloat16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('bf16 pmode $V', d['ms_per_step'], 'gate_fwd', d['kernel_ms']['gate_fwd'])"
done | tee $OUT/bench_bf16.txt
unset MADELEINE_BF16_GATE_PERSIST
for V in 1 2; do
MADELEINE_GATE_PERSIST=$V MADELEINE_BF16_GATE_PERSIST=$V timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_split_gpu.py tests/test_bf16_gpu.py -m gpu -x -q -k "gate" 2>&1 | tail -2
done | tee $OUT/tests.txt
