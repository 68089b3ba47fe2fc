#!/bin/bash
# sclk / power while a kernel family runs in a loop: is the split engine clock/power limited?
cd "$GRAFT_REPO_ROOT" || exit 1
sample() { for i in 1 2 3 4 5 6; do sleep 0.7; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket" | tr '\n' ' '; echo; done; }
echo "== split lab (fp16x2 + bf16x3 NT loops)"; (timeout 60 tools/micro/build/split_lab 400 1024 512 > /tmp/lab.log 2>&1 &) ; sleep 1.5; sample; wait; sleep 2
echo "== fp32 MFMA bench step loop"; (MADELEINE_GEMM=fp32 timeout 120 python bench.py --steps 120 --warmup 3 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg > /tmp/b1.log 2>&1 &); sleep 12; sample; sleep 3
echo "== split bench step loop"; (MADELEINE_GEMM=split timeout 120 python bench.py --steps 250 --warmup 3 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg > /tmp/b2.log 2>&1 &); sleep 12; sample
