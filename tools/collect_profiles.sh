#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 passes over bench.py / tools/prof_kernels.py / tools/bench_got.py, summarised to text
# under gpurun_out/prof_txt/ (the rocpd .db files are too big to travel back).  Usage: bash tools/collect_profiles.sh <tag> [what]
# PMC passes use --kernel-trace only (gpurun refuses --pmc combined with other trace domains).
set -u
TAG=${1:-r02}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_txt
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, summary-top-n, rocprof args..., -- command
    local name=$1 top=$2; shift 2
    rm -rf /tmp/prof_$name
    rocprofv3 "$@" > /tmp/prof_$name.log 2>&1
    { echo "# $TAG $name: rocprofv3 $*" | sed "s#$R/##g"; python $R/tools/rocpd_summary.py /tmp/prof_$name/*/*.db $top; } > $OUT/${TAG}_$name.txt 2>&1
    rm -rf /tmp/prof_$name
}
B="--no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
if [ $WHAT = all ] || [ $WHAT = bench ]; then
run bench_c2_kernel_stats 40 --kernel-trace --stats -d /tmp/prof_bench_c2_kernel_stats -- python $R/bench.py --steps 3 --warmup 1 $B
run bench_c2_bf16_kernel_stats 45 --kernel-trace --stats -d /tmp/prof_bench_c2_bf16_kernel_stats -- python $R/bench.py --steps 3 --warmup 1 $B --precision bfloat16
run bench_c3_kernel_stats 50 --kernel-trace --stats -d /tmp/prof_bench_c3_kernel_stats -- python $R/bench.py --config c3 --steps 3 --warmup 1 $B
fi
if [ $WHAT = all ] || [ $WHAT = kernels ]; then
run kernels_c2_stats 25 --kernel-trace --stats -d /tmp/prof_kernels_c2_stats -- python $R/tools/prof_kernels.py --iters 5
run pool_pmc_fetch 8 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_pool_pmc_fetch -- python $R/tools/prof_kernels.py --iters 2 --only pool
run pool_pmc_write 8 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_pool_pmc_write -- python $R/tools/prof_kernels.py --iters 2 --only pool
run gate_pmc_sq 40 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d /tmp/prof_gate_pmc_sq -- python $R/tools/prof_kernels.py --iters 1 --only gate
run gate_pmc_tcc 40 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_gate_pmc_tcc -- python $R/tools/prof_kernels.py --iters 1 --only gate
fi
if [ $WHAT = all ] || [ $WHAT = got ]; then
# GOT kernels alone (tools/bench_got.py: k = 32 cases, n = 32 .. 256 tokens, fwd + bwd): durations, then SQ / memory counters
run got_kernel_stats 40 --kernel-trace --stats -d /tmp/prof_got_kernel_stats -- python $R/tools/bench_got.py
run got_pmc_sq 40 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d /tmp/prof_got_pmc_sq -- python $R/tools/bench_got.py
run got_pmc_fetch 40 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace -d /tmp/prof_got_pmc_fetch -- python $R/tools/bench_got.py
run got_pmc_write 40 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_got_pmc_write -- python $R/tools/bench_got.py
fi
if [ $WHAT = split ]; then
# Counters of the split-engine matrix-core kernels at config-2 geometry (tools/prof_split.py: gate fwd / dz / dX / dW, Linear 512 -> 2048
# fwd / dX / dW), one --pmc pass per counter group; tools/pmc_mfma_busy.py turns the databases into per-kernel MFMA-busy fractions
PS="python $R/tools/prof_split.py --iters 2"
pmc() {  # name, counters...
    local name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -- $PS > /tmp/prof_$name.log 2>&1
    { echo "# $TAG $name: rocprofv3 --pmc $* --kernel-trace -- $PS" | sed "s#$R/##g"; python $R/tools/rocpd_summary.py /tmp/prof_$name/*/*.db 60; } > $OUT/${TAG}_$name.txt 2>&1
}
pmc split_pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc split_pmc_insts SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
pmc split_pmc_wait SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
pmc split_pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum
python $R/tools/pmc_mfma_busy.py $OUT/${TAG}_split_mfma_busy.json /tmp/prof_split_pmc_*/*/*.db > $OUT/${TAG}_split_mfma_busy.txt 2>&1
rm -rf /tmp/prof_split_pmc_*
fi
ls -la $OUT
