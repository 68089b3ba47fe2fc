B="python bench.py --config c5 --precision bfloat16 --steps 4 --warmup 2 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in "3 31" "2 2" "3 2" "2 31" "3 31" "2 2"; do
  set -- $q
  MADELEINE_BF16_STAGES=$1 MADELEINE_BF16_LIN_STAGES=$2 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5BF16 stages', '$1', 'lin', '$2', d['ms_per_step'], d.get('kernels'))" >> gpurun_out/r06u_c5_bf16_stages.txt
done
cat gpurun_out/r06u_c5_bf16_stages.txt
