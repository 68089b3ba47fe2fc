#!/bin/bash
# stain-encoding concat folded into the first Linear (group bias): tests, then config 5 step before / after is in profiles (134.9 ms round 4)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05m}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_path_gpu.py tests/test_bf16_gpu.py -m gpu -x -q -k "stain or ragged or c5 or golden or encoder" > $OUT/pytest_stain.log 2>&1; echo "stain tests rc $?"; tail -4 $OUT/pytest_stain.log
timeout 600 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg > $OUT/c5.json 2> $OUT/c5.err; tail -c 300 $OUT/c5.err
python - <<PY
import json
d=json.loads(open("$OUT/c5.json").read().strip().splitlines()[-1]); print("c5", d["ms_per_step"], d["value"]); print(d["kernel_ms"]); print(d["kernel_calls_per_step"])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
