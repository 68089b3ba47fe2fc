#!/bin/bash
# bf16 LN kernels after the polynomial GELU: row geometry variants (fwd <4,2>, bwd <2,4>) vs shipped (<8,1>, <4,2>); then all GPU tests
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05y}; mkdir -p $OUT
cd $R
for V in default lnf42 lnb24 default lnf42 lnb24; do
  if [ $V = default ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  echo "== $V"; timeout 200 python tools/exp_ln.py 2>/dev/null | grep "W=2048"
done | tee $OUT/exp_ln.txt
unset MADELEINE_LIB
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
