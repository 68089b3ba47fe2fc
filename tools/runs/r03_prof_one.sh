#!/bin/bash
# usage: r03_prof_one.sh <tag> <bench args...>   (MADELEINE_GEMM from the environment)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
rm -rf /tmp/prof_one
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_one -- python $R/bench.py --steps 3 --warmup 1 $B "$@" > /tmp/prof_one.log 2>&1
{ echo "# $TAG: MADELEINE_GEMM=${MADELEINE_GEMM:-split} rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 $B $*   (clock: under the profiler)"; python $R/tools/rocpd_summary.py /tmp/prof_one/*/*.db 50; } > $OUT/$TAG.txt 2>&1
sed -n 3,34p $OUT/$TAG.txt | awk '{printf "%-64s %5s %9s %9s %6s\n", substr($1,length($1)-62), $2, $3, $6, $7}'
tail -1 $OUT/$TAG.txt
