#!/bin/bash
# dz passes with the dropout mode fixed at compile time (and coalesced float4 loads in the split one) vs HEAD~ (tools/ab/head.so), same box; then all GPU tests
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05r}; mkdir -p $OUT
cd $R
for P in float32 bfloat16; do
for V in head default head default; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg --precision $P 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms']; print('$P $V', d['ms_per_step'], {n: k[n] for n in ('gate_bwd_dz','gate_fwd','ln_gelu_drop_fwd','ln_gelu_drop_bwd')})"
done; done | tee $OUT/bench.txt
unset MADELEINE_LIB
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
