#!/bin/bash
# rocprofv3 kernel stats of the config-3 step (5 stains + GOT): where the non-mdl (torch glue) kernels are
TAG=${1:-r05c3}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c3
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -- python $R/bench.py --config c3 --steps 3 --warmup 1 $B > /tmp/prof_c3.log 2>&1
{ echo "# $TAG bench_c3_kernel_stats: BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats -- python bench.py --config c3 --steps 3 --warmup 1 $B (4 steps in the trace)"; python $R/tools/rocpd_summary.py /tmp/prof_c3/*/*.db 90; } > $OUT/${TAG}_bench_c3_kernel_stats.txt 2>&1
grep -v "_ZN3mdl" $OUT/${TAG}_bench_c3_kernel_stats.txt | head -60 | cut -c1-150
# the loss section of one step: from the token projector's tall product to the pooling backward
for db in /tmp/prof_c3/*/*.db; do python $R/tools/rocpd_timeline.py $db sp_nt_tall 4 260 2 > $OUT/${TAG}_bench_c3_loss_section_sequence.txt 2>&1; done
