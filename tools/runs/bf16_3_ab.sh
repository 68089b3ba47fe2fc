python -m pytest tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs --precision bfloat16"
for q in 2 3 2 3; do
  MADELEINE_BF16_STAGES=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BF16STAGES', $q, d['ms_per_step'], d['kernels'])" >> gpurun_out/r06r_bf16_spread.txt
done
cat gpurun_out/r06r_bf16_spread.txt
