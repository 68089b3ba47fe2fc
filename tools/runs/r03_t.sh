#!/bin/bash
# round 3: full GPU test suite + a default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_tests.log
tail -15 gpurun_out/r03_tests.log
if [ -z "$NO_BENCH" ]; then
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -c 3000 gpurun_out/r03_bench.json
fi
