cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/t.log 2>&1
tail -15 gpurun_out/t.log
if [ -n "$2" ]; then python tools/exp_linear.py > gpurun_out/exp_linear.log 2>&1; cat gpurun_out/exp_linear.log; fi
