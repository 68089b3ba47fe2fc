#!/bin/bash
# generic dz pass (bf16 / fp32-MFMA modes): all heads per workgroup, rows in flight per thread (MDL_DZ_UNROLL 2 | 4) vs HEAD~, same box; bf16 + gate tests
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05s}; mkdir -p $OUT
cd $R
for V in head dzu2 default head dzu2 default; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg --precision bfloat16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms']; print('bf16 $V', d['ms_per_step'], {n: k[n] for n in ('gate_bwd_dz','gate_fwd')})"
done | tee $OUT/bench.txt
unset MADELEINE_LIB
timeout 1200 python -m pytest tests/test_bf16_gpu.py tests/test_hip_kernels.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
