#!/bin/bash
# same-box A/B of two prebuilt libraries (tools/micro/build/lib_old.so / lib_new.so): c2 lines, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in old new; do
  cp tools/micro/build/lib_$v.so madeleine_amd/csrc/libmadeleine_amd.so; touch madeleine_amd/csrc/libmadeleine_amd.so
  echo -n "$v: "; timeout 300 bash tools/runs/r03_c3.sh ${1:-c2:float32} | cut -c 1-${2:-330}
done
done
