B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs --precision bfloat16"
for q in 2 31 32 2 31 32; do
  MADELEINE_BF16_LIN_STAGES=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('BF16LIN', $q, d['ms_per_step'], 'linear_bwd', k['linear_bwd'][0], 'linear_fwd', k['linear_fwd'][0], 'gate_bwd_gemm', k['gate_bwd_gemm'][0])" >> gpurun_out/r06s_bf16_lin_stages.txt
done
cat gpurun_out/r06s_bf16_lin_stages.txt
