python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06o_gputest.log; cat gpurun_out/r06o_gputest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in 2 3 2 3; do
  MADELEINE_SP_NT_STAGES=$q $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NTSTAGES', $q, d['ms_per_step'], d['kernels'])" >> gpurun_out/r06o_nt_spread_all.txt
done
for q in 2 3; do
  MADELEINE_SP_NT_STAGES=$q $B --config c3 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 NTSTAGES', $q, d['ms_per_step'], d['kernels'])" >> gpurun_out/r06o_nt_spread_all.txt
done
cat gpurun_out/r06o_nt_spread_all.txt
