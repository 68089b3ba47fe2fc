cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof_txt
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-extra-legs $2 > /tmp/prof_b.log 2>&1
tail -2 /tmp/prof_b.log | cut -c1-300
{ echo "# $1: rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-extra-legs $2"; python $R/tools/rocpd_summary.py /tmp/prof_b/*/*.db 60; } > $R/gpurun_out/prof_txt/$1_kernel_stats.txt 2>&1
cat $R/gpurun_out/prof_txt/$1_kernel_stats.txt | cut -c1-160 | head -64
