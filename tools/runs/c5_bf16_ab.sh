B="python bench.py --config c5 --precision bfloat16 --steps 4 --warmup 2 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in 32768 4096 32768 4096; do
  MADELEINE_SPLIT_TOKENS=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5BF16', $q, d['ms_per_step'], d.get('kernels'))" >> gpurun_out/r06g_c5_bf16_ab.txt
done
B="python bench.py --config c5 --steps 3 --warmup 2 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in 32768 4096; do
  MADELEINE_SPLIT_TOKENS=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5F32', $q, d['ms_per_step'], d.get('kernels'))" >> gpurun_out/r06g_c5_bf16_ab.txt
done
B="python bench.py --config c3 --steps 4 --warmup 2 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in 32768 4096; do
  MADELEINE_SPLIT_TOKENS=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3F32', $q, d['ms_per_step'], d.get('kernels'))" >> gpurun_out/r06g_c5_bf16_ab.txt
done
cat gpurun_out/r06g_c5_bf16_ab.txt
