#!/bin/bash
# c3 / c5 / bf16 single lines (no secondary legs)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in "$@"; do
  timeout 900 python bench.py --config ${cfg%%:*} --precision ${cfg##*:} --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | tee gpurun_out/r03_line_${cfg%%:*}_${cfg##*:}.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], json.dumps(d['kernel_ms']))"
done
