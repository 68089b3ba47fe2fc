#!/bin/bash
# rocprofv3 kernel stats of the c2 step (GEMM mode from $MADELEINE_GEMM), tag $1
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03b}
OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
name=bench_c2_${MADELEINE_GEMM:-split}_kernel_stats
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $R/bench.py --steps 3 --warmup 1 $B > /tmp/prof_$name.log 2>&1
{ echo "# $TAG $name: MADELEINE_GEMM=${MADELEINE_GEMM:-split} rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 $B   (profiler clocks: durations 7-12 % above bench.py's event times)"; python $R/tools/rocpd_summary.py /tmp/prof_$name/*/*.db 60; } > $OUT/${TAG}_$name.txt 2>&1
cut -c 30-200 $OUT/${TAG}_$name.txt | head -50
