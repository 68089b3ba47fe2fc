#!/bin/bash
# A/B of several tools/ab/<name>.so against the in-tree library on one box (config-2 step, two rounds)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04ab; mkdir -p $OUT
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg"
for rep in 1 2; do
 for V in base "$@"; do
  if [ $V = base ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 200 python $R/bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('$V $rep: step %.3f | dz %.3f gate_fwd %.3f gate_bwd_gemm %.3f ln_fwd %.3f ln_bwd %.3f' % (d['ms_per_step'], k['gate_bwd_dz'], k['gate_fwd'], k['gate_bwd_gemm'], k['ln_gelu_drop_fwd'], k['ln_gelu_drop_bwd']))"
 done
done
