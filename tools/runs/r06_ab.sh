#!/bin/bash
# A/B of the in-tree library against tools/ab/<name>.so variants on one box: the config-2 step, alternating, twice.  Usage: r06_ab.sh name...
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06ab; mkdir -p $OUT
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg"
for rep in 1 2; do
 for V in "$@" base; do
  if [ $V = base ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 200 python $R/bench.py $B ${BENCH_EXTRA:-} 2>/dev/null | tail -1 > $OUT/${V}_$rep.json
  python - <<PY
import json
d=json.loads(open("$OUT/${V}_$rep.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("$V $rep: step %.3f | gate_fwd %.3f gate_bwd_gemm %.3f linear_fwd %.3f linear_bwd %.3f" % (d["ms_per_step"], k["gate_fwd"][0], k["gate_bwd_gemm"][0], k["linear_fwd"][0], k["linear_bwd"][0]))
PY
 done
done
