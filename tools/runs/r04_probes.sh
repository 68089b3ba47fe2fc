#!/bin/bash
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for V in base nodma nowait nobar nosync none; do
  if [ $V = base ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 120 python $R/tools/exp_loop_probes.py $V 2>/dev/null | tail -1
done
done
