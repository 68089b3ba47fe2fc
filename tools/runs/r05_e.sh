#!/bin/bash
# Round 5: new tests (any patch_embedding_dim, bf16 per-parameter gradients + curve, batched inference) then the whole GPU suite, then a full bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05e}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bf16_gpu.py -m gpu -x -q -s -k "any_patch or autocast_parameter or short_training or inference_full_bag" > $OUT/pytest_new.log 2>&1; echo "new tests rc $?"; grep -E "passed|failed|HIP bf16|loss:|max relative|Error|assert" $OUT/pytest_new.log | head -120
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-pmc > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"])
for k in ("grad_terms2_mode","bf16_mode","fp32_mfma_mode","host_input_mode","c3_mode","c4_rank_emulation","inference_mode","c3_all_present","c4_rank_emulation_all_present","c5_rank_emulation"):
    if k in d: print(k, d[k].get("ms_per_step"), d[k].get("value"), d[k].get("got_ms_per_step_sum_over_stains"), d[k].get("implied_weak_scaling_ceiling_vs_single_rank"), d[k].get("implied_weak_scaling_ceiling_vs_c3_single_rank"))
print(json.dumps(d["inference_mode"]))
PY
