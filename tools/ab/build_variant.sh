#!/bin/bash
# Builds a variant of libmadeleine_amd.so with extra compiler flags into tools/ab/<name>.so (for MADELEINE_LIB=... A/B runs on one box).
# Usage: tools/ab/build_variant.sh <name> <flags...>      e.g.  build_variant.sh prio -DMDL_SP_SETPRIO
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/../.. && pwd)
OBJ=$(mktemp -d)
for f in $R/madeleine_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o $OBJ/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $R/tools/ab/$NAME.so
rm -rf $OBJ
ls -la $R/tools/ab/$NAME.so
