#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05g}; mkdir -p $OUT
cd $R
for v in 0 1; do
  if [ $v = 1 ]; then export MADELEINE_GOT_NO512=1; else unset MADELEINE_GOT_NO512; fi
  echo "NO512=$v"; GOT_GEOMS='[[32,256],[128,256],[32,192]]' timeout 300 python tools/bench_got.py 2>/dev/null | grep "k="
done
unset MADELEINE_GOT_NO512
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "got" 2>&1 | tail -2
