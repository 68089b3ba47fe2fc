set -x
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in 4096 16384 32768 4096 32768; do
  MADELEINE_SPLIT_TOKENS=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SPLITQ fp32', $q, d['ms_per_step'], d['kernels'])" >> gpurun_out/r06e_split.txt
done
for q in 4096 16384 32768 4096 32768; do
  MADELEINE_SPLIT_TOKENS=$q $B --precision bfloat16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SPLITQ bf16', $q, d['ms_per_step'], d['kernels'])" >> gpurun_out/r06e_split.txt
done
python tools/launch_tail.py > gpurun_out/r06e_launch_tail.txt 2>&1
cat gpurun_out/r06e_split.txt
tail -60 gpurun_out/r06e_launch_tail.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
MADELEINE_BF16_GATE128=1 $B --precision bfloat16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GATE128 bf16', d['ms_per_step'], d['kernels'])" >> gpurun_out/r06e_split.txt
tail -1 gpurun_out/r06e_split.txt
