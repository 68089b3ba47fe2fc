#!/bin/bash
# GOT after the one-pass reverse sweep: parity tests, timings by size class, four-stain batch at n = 256
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05got}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_bench_path_gpu.py -m gpu -x -q -k "got" > $OUT/pytest_got.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_got.log
GOT_GEOMS='[[32,32],[32,64],[32,128],[25,188],[32,192],[32,256],[128,256]]' timeout 300 python tools/bench_got.py > $OUT/bench_got.txt 2>&1; cat $OUT/bench_got.txt
