# non-default A/B switches still pass their tests (two-stage loops, exact-fp32 engine, bf16 Linear stage variants, old split counts)
T="tests/test_split_gpu.py tests/test_hip_kernels.py tests/test_bf16_gpu.py tests/test_model_gpu.py"
echo "== two-stage loops"; MADELEINE_SP_NT_STAGES=2 MADELEINE_SP_TN_STAGES=2 MADELEINE_BF16_STAGES=2 MADELEINE_BF16_LIN_STAGES=2 python -m pytest $T -m gpu -x -q 2>&1 | tail -1
echo "== bf16 Linears three-stage"; MADELEINE_BF16_LIN_STAGES=3 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -1
echo "== 4096-token splits, GOT whole products"; MADELEINE_SPLIT_TOKENS=4096 MADELEINE_GOT_NO_HALF_PRODUCTS=1 python -m pytest tests/test_split_gpu.py tests/test_bf16_gpu.py tests/test_got_rank_shapes_gpu.py -m gpu -x -q 2>&1 | tail -1
echo "== exact-fp32 engine"; MADELEINE_GEMM=fp32 python -m pytest tests/test_model_gpu.py tests/test_hip_kernels.py -m gpu -x -q 2>&1 | tail -1
