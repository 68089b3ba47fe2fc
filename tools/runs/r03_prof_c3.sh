#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
name=bench_c3_kernel_stats
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $R/bench.py --config c3 --steps 3 --warmup 1 $B > /tmp/prof_$name.log 2>&1
{ echo "# r03a $name: rocprofv3 --kernel-trace --stats -- python bench.py --config c3 --steps 3 --warmup 1 $B   (profiler clocks: durations 7-12 % above bench.py's event times)"; python $R/tools/rocpd_summary.py /tmp/prof_$name/*/*.db 60; } > $OUT/r03a_$name.txt 2>&1
cat $OUT/r03a_$name.txt | cut -c 40-200 | head -45
