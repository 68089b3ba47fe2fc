#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" ; 
MADELEINE_GEMM=split timeout 170 python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg > /tmp/b.log 2>&1 &
PID=$!
sleep 20
for i in $(seq 1 12); do sleep 0.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '; echo; done
wait $PID; tail -c 3000 /tmp/b.log | grep -o '"value": [0-9.]*, "unit": "slides/s", "n_gpus": 1, "steps": [0-9]*, "warmup": 3, "ms_per_step": [0-9.]*'; true
