MADELEINE_SP_TN_STAGES=3 python -m pytest tests/test_split_gpu.py tests/test_split_range_gpu.py tests/test_grad_terms_gpu.py tests/test_hip_kernels.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in 2 3 2 3; do
  MADELEINE_SP_TN_STAGES=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TNSTAGES', $q, d['ms_per_step'], d['kernels'])" >> gpurun_out/r06p_tn_spread.txt
done
cat gpurun_out/r06p_tn_spread.txt
