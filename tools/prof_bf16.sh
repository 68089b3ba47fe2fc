set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bf
rocprofv3 --kernel-trace --stats -d /tmp/prof_bf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-leg --precision bfloat16 > /tmp/prof_bf.log 2>&1
{ echo "# r01d bench_c2_bf16_kernel_stats: rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-leg --precision bfloat16"; python $R/tools/rocpd_summary.py /tmp/prof_bf/*/*.db 45; } > $OUT/r01d_bench_c2_bf16_kernel_stats.txt 2>&1
tail -1 /tmp/prof_bf.log | cut -c1-300
