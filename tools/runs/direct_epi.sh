python -m pytest tests/test_split_gpu.py tests/test_split_range_gpu.py tests/test_grad_terms_gpu.py -m gpu -x -q 2>&1 | tail -2
python tools/exp_nt_shapes.py 2>&1 | grep "^M" | head -7
MADELEINE_LIB=$GRAFT_REPO_ROOT/tools/ab/ntT.so python tools/exp_nt_shapes.py 2>&1 | grep "^M" | head -7 | sed 's/^/old-epilogue /'
bash tools/runs/r06_ab.sh ntT
