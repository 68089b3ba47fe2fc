#!/bin/bash
# LayerNorm-GELU-Dropout kernels at 2048 wide: (NV, WPR) and grid-cap variants, fp32 and bf16 storage (tools/exp_ln.py)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05i}; mkdir -p $OUT
cd $R
for V in base lnf42 lnf24 lnb81 lnb24 lncap768 lncap1024 base; do
  if [ $V = base ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  echo "== $V"; timeout 120 python tools/exp_ln.py 2>/dev/null | grep "W=2048"
done | tee $OUT/ln_variants.txt
