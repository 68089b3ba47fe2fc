#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for mode in fp32 split; do
  MADELEINE_GEMM=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['value'], d['ms_per_step'], json.dumps(d['kernel_ms'])); print(json.dumps({k:(v['achieved'],v['unit']) for k,v in d['kernel_roofline'].items()}))"
done
