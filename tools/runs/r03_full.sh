#!/bin/bash
# round 3: default bench line (all legs) + rocprofv3 kernel stats of the c2 step in both GEMM modes, the bf16 mode and c3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof_txt
timeout 1500 python bench.py > gpurun_out/r03i_bench_default.json 2> gpurun_out/r03i_bench_default.err
tail -c 400 gpurun_out/r03i_bench_default.err
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_txt
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
prof() {  # name, extra env, args
  local name=$1; shift
  rm -rf /tmp/prof_$name
  env "$1" rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $R/bench.py --steps 3 --warmup 1 $B ${@:2} > /tmp/prof_$name.log 2>&1
  { echo "# r03i $name: $1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 $B ${@:2}   (clock: under the profiler; bench.py's HIP-event times are 5-12 % shorter)"; python $R/tools/rocpd_summary.py /tmp/prof_$name/*/*.db 45; } > $OUT/r03i_$name.txt 2>&1
  rm -rf /tmp/prof_$name
}
prof bench_c2_split_kernel_stats MADELEINE_GEMM=split
prof bench_c2_fp32mfma_kernel_stats MADELEINE_GEMM=fp32
prof bench_c2_bf16_kernel_stats MADELEINE_GEMM=split --precision bfloat16
prof bench_c3_split_kernel_stats MADELEINE_GEMM=split --config c3
ls -la $OUT | tail -6
