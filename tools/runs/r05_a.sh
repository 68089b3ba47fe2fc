#!/bin/bash
# Round-5 first pass: GPU test suite, HBM access-mix sweep, baseline bench line (no CPU baseline / PMC: measured later)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05a; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
timeout 300 tools/micro/build/hbm_rate > $OUT/hbm_rate.txt 2>&1; tail -20 $OUT/hbm_rate.txt
timeout 900 python bench.py --no-cpu-baseline --no-pmc > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"])
for k in ("grad_terms2_mode","bf16_mode","fp32_mfma_mode","host_input_mode","c3_mode","c4_rank_emulation","inference_mode","c3_all_present","c4_rank_emulation_all_present","c5_rank_emulation"):
    if k in d: print(k, d[k].get("ms_per_step"), d[k].get("value"), d[k].get("got_ms_per_step_sum_over_stains"), d[k].get("implied_weak_scaling_ceiling_vs_single_rank"))
print(json.dumps(d["kernel_ms"]))
PY
