#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench step (c2, split mode); summary -> gpurun_out/prof_txt/<tag>_bench_c2_split_kernel_stats.txt
TAG=${1:-r04a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 1 $B > /tmp/prof_c2.log 2>&1
{ echo "# $TAG bench_c2_split_kernel_stats: BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 $B (4 steps in the trace)"; python $R/tools/rocpd_summary.py /tmp/prof_c2/*/*.db 70; } > $OUT/${TAG}_bench_c2_split_kernel_stats.txt 2>&1
for db in /tmp/prof_c2/*/*.db; do python $R/tools/rocpd_gaps.py $db FusedAdam 4 6 20 > $OUT/${TAG}_bench_c2_device_gaps.txt 2>&1; done
head -75 $OUT/${TAG}_bench_c2_split_kernel_stats.txt | cut -c1-150; head -30 $OUT/${TAG}_bench_c2_device_gaps.txt
for db in /tmp/prof_c2/*/*.db; do python $R/tools/rocpd_timeline.py $db FusedAdam 172 2 5 > $OUT/${TAG}_bench_c2_step_sequence.txt 2>&1; done
