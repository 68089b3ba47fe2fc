# round-6 closing run: GPU suite, default bench (compact line + detail), rocprofv3 kernel stats of the bench
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06f_gputest.log
python bench.py > gpurun_out/r06f_bench.log 2>&1
tail -1 gpurun_out/r06f_bench.log > gpurun_out/r06f_bench_compact_line.json
cp gpurun_out/bench_detail.json gpurun_out/r06f_bench_detail.json
bash tools/collect_profiles.sh r06f bench > /dev/null 2>&1
cat gpurun_out/r06f_gputest.log
tail -c 1500 gpurun_out/r06f_bench_compact_line.json
