#!/bin/bash
# split-mode LayerNorm kernels (fp32 in, image out): hash dropout mode fixed at compile time (tools/ab/lnimgdm.so) vs run-time mode (default); c2 step
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05z}; mkdir -p $OUT
cd $R
for V in default lnimgdm default lnimgdm default lnimgdm; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms']; print('$V', d['ms_per_step'], {n: k[n] for n in ('ln_gelu_drop_fwd','ln_gelu_drop_bwd')}, d['config'].get('final_loss'))"
done | tee $OUT/bench.txt
