#!/bin/bash
# whole-line image stores through the lane-pair exchange (sp_img_store4): dz pass + LayerNorm image outputs, vs round-4 store patterns (same box)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05q}; mkdir -p $OUT
cd $R
for V in head default nopair head default nopair; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms']; print('$V', d['ms_per_step'], {n: k[n] for n in ('gate_bwd_dz','ln_gelu_drop_fwd','ln_gelu_drop_bwd','split_image','pool_fwd')})"
done | tee $OUT/bench_split.txt
unset MADELEINE_LIB
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
