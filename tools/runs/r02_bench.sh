cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps ${1:-20} --warmup 5 ${2} > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/t_all.log 2>&1
tail -4 gpurun_out/t_all.log
