#!/bin/bash
# Round-5 evidence pass: default bench line (all legs, CPU baseline, in-run PMC), rocprofv3 kernel stats of the headline step, GOT kernel stats
TAG=${1:-r05k}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
SECONDS=0; timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench.py default run: $SECONDS s wall"; tail -c 300 $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
for k in ("grad_terms2_mode","bf16_mode","fp32_mfma_mode","host_input_mode","c3_mode","c4_rank_emulation","inference_mode","c3_all_present","c4_rank_emulation_all_present","c5_rank_emulation"):
    if k in d: print(k, d[k].get("ms_per_step"), d[k].get("value"), d[k].get("got_ms_per_step_sum_over_stains"), d[k].get("implied_weak_scaling_ceiling_vs_single_rank"), d[k].get("implied_weak_scaling_ceiling_vs_c3_single_rank"))
print(d.get("power_clock")); print(d["headline_summary"])
PY
bash tools/runs/r04_prof_c2.sh $TAG > $OUT/prof_c2.log 2>&1; tail -5 $OUT/prof_c2.log
bash tools/collect_profiles.sh $TAG got > /dev/null 2>&1
# bf16 mode: kernel stats of the same step under torch.autocast(bfloat16)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c2b && BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs --precision bfloat16 > /tmp/prof_c2b.log 2>&1
  { echo "# $TAG bench_c2_bf16_kernel_stats: BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs --precision bfloat16"; python $R/tools/rocpd_summary.py /tmp/prof_c2b/*/*.db 45; } > $R/gpurun_out/prof_txt/${TAG}_bench_c2_bf16_kernel_stats.txt 2>&1 )
ls $R/gpurun_out/prof_txt | grep $TAG
