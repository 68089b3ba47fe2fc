#!/bin/bash
# A/B of the in-tree library against tools/ab/<name>.so on one box: the config-2 step, alternating, twice
NAME=${1:-prio}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04ab; mkdir -p $OUT
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg"
for rep in 1 2; do
 for V in $NAME base; do
  if [ $V = base ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
  timeout 200 python $R/bench.py $B > $OUT/${V}_$rep.json 2>/dev/null
  python - <<EOF
import json
d=json.loads(open("$OUT/${V}_$rep.json").read().strip().splitlines()[-1]); k=d["kernel_ms"]
print("$V $rep: step %.3f | gate_fwd %.3f gate_bwd_gemm %.3f linear_fwd %.3f linear_bwd %.3f" % (d["ms_per_step"], k["gate_fwd"], k["gate_bwd_gemm"], k["linear_fwd"], k["linear_bwd"]))
EOF
 done
done
