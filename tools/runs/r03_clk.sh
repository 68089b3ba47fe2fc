#!/bin/bash
# sclk / socket power while the bench step loop runs (is the step clock/power limited?)
cd "$GRAFT_REPO_ROOT" || exit 1
sample() { for i in $(seq 1 $1); do sleep 0.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '; echo; done; }
for mode in fp32 split split; do
  echo "== bench c2 step loop, MADELEINE_GEMM=$mode"
  MADELEINE_GEMM=$mode timeout 170 python bench.py --steps 600 --warmup 3 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg > /tmp/b_$mode.log 2>&1 &
  PID=$!
  sleep 22; sample 10
  wait $PID; tail -c 300 /tmp/b_$mode.log | grep -o '"value": [0-9.]*, "unit": "slides/s", "n_gpus": 1, "steps": [0-9]*' 
done
