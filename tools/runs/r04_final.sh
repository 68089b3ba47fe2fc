#!/bin/bash
# Round-4 evidence pass: default bench line, accuracy report, rocprofv3 summaries (kernel stats / PMC) -> gpurun_out/r04f, gpurun_out/prof_txt
TAG=${1:-r04f}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.err
timeout 600 python tools/split_range_report.py $OUT/split_range_report.json > /dev/null 2> $OUT/split_range_report.err
bash tools/runs/r04_prof_c2.sh $TAG > /dev/null 2>&1
bash tools/collect_profiles.sh $TAG all > /dev/null 2>&1
ls -la $R/gpurun_out/prof_txt | grep $TAG
python - <<EOF
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
for k in ("grad_terms2_mode","bf16_mode","fp32_mfma_mode","host_input_mode","c3_mode","c4_rank_emulation","inference_mode"):
    if k in d: print(k, d[k].get("ms_per_step"), d[k].get("value"))
print(d.get("power_clock"))
EOF
