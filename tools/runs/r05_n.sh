#!/bin/bash
# Pooling forward: early row requests (MDL_POOL_PRE) and occupancy target A/B, then the pooling / reproducibility tests on the default build
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05n}; mkdir -p $OUT
cd $R
for V in pool_base default pool_base8 pool_pre8 pool_base default; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  echo "== $V"; timeout 200 python tools/exp_pool.py 2>&1 | grep -E "pool_|Error|error"
done | tee $OUT/pool_variants.txt
unset MADELEINE_LIB
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_model_gpu.py -m gpu -x -q -k "pool or reproducible or ragged or view or bit" 2>&1 | tail -5 | tee $OUT/tests.txt
