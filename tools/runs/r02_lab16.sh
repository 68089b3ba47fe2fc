cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tools/micro/build/gemm_lab_bf16 ${1:-8} > gpurun_out/lab16.log 2>&1
cat gpurun_out/lab16.log
