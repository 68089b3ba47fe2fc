#!/bin/bash
# Pooling forward: occupancy target (waves per SIMD 2 / 4 / default 6) A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05n}; mkdir -p $OUT
cd $R
for V in default pool_wpe2 pool_w2u4 pool_u4 pool_w2u2 pool_w4u4 default pool_wpe2 pool_w2u4 pool_u4 pool_w2u2 pool_w4u4 default pool_wpe2 pool_w2u4 pool_u4 pool_w2u2 pool_w4u4; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  echo "== $V"; timeout 200 python tools/exp_pool.py 2>&1 | grep -E "pool_|Error|error"
done | tee $OUT/pool_variants.txt
