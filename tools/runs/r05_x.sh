#!/bin/bash
# bf16 LayerNorm-GELU-Dropout kernels: polynomial GELU + compile-time dropout mode vs HEAD (tools/ab/head.so), same box; bf16 tests
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05x}; mkdir -p $OUT
cd $R
for V in head default head default; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg --precision bfloat16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms']; r=d['kernel_roofline']; print('bf16 $V', d['ms_per_step'], {n: k[n] for n in ('ln_gelu_drop_fwd','ln_gelu_drop_bwd')}, r['ln_gelu_drop_fwd']['achieved'], r['ln_gelu_drop_bwd']['achieved'], d['config'].get('final_loss'))"
done | tee $OUT/bench.txt
for V in head default; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  echo "== $V"; timeout 200 python tools/exp_ln.py 2>/dev/null | grep -i "bf16\|W=2048" | head -12
done | tee $OUT/exp_ln.txt
unset MADELEINE_LIB
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_hip_kernels.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
