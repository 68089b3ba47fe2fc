B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-pmc --no-extra-legs"
for q in "2 0" "2 1" "2 2" "3 0" "2 0" "2 2"; do
  set -- $q
  MADELEINE_SP_NT_STAGES=$1 MADELEINE_GATE_PERSIST=$2 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('NTSTAGES $1 PERSIST $2', d['ms_per_step'], 'gate_fwd', k['gate_fwd'][0])" >> gpurun_out/r06z_persist.txt
done
cat gpurun_out/r06z_persist.txt
