cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof_txt
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-leg $2 > /tmp/prof_b.log 2>&1
{ echo "# $1 bench_c2_kernel_stats: rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-leg $2"; python $R/tools/rocpd_summary.py /tmp/prof_b/*/*.db 45; } > $R/gpurun_out/prof_txt/$1_kernel_stats.txt 2>&1
cat $R/gpurun_out/prof_txt/$1_kernel_stats.txt | cut -c1-150 | head -40
