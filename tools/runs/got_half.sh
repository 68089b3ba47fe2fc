python tools/runs/got_half_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06l_got_half_products_ab.txt
python -m pytest tests/test_got_rank_shapes_gpu.py tests/test_hip_kernels.py tests/test_bench_path_gpu.py -m gpu -x -q -k "got or GOT or void or c4 or rank" 2>&1 | tail -4
