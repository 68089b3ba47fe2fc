#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
BENCH_NO_TIMER=1 MADELEINE_PIPELINE_CHUNKS=1 run "chunks1 notimer"
BENCH_NO_TIMER=1 MADELEINE_PIPELINE_CHUNKS=2 run "chunks2 notimer"
BENCH_NO_TIMER=1 MADELEINE_PIPELINE_CHUNKS=2 MADELEINE_PIPELINE_SERIAL=1 run "chunks2 serial notimer"
