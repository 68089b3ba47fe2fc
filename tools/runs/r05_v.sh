#!/bin/bash
# bf16 Linear NT on the 256-tile: persistent over column tiles (MADELEINE_BF16_LIN_PERSIST 0|1) x smallest contraction on that tile (1024 | 512); per-product times
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05v}; mkdir -p $OUT
cd $R
for rep in 1 2; do
for V in "0 1024" "1 1024" "0 512" "1 512"; do
  set -- $V
  export MADELEINE_BF16_LIN_PERSIST=$1 MADELEINE_BF16_LIN256_MINK=$2
  echo "== persist $1 mink $2"; timeout 300 python tools/exp_linear_bf16.py 2>/dev/null
done; done | tee $OUT/linear.txt
