#!/bin/bash
# Round 5: (1) contiguous vs grid-stride row order in the LayerNorm / image passes; (2) dW products on a side stream (+ CU masks)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05d}; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_bench_path_gpu.py -m gpu -x -q -k "reproducible" 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg"
show() {
python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1]); k=d["kernel_ms"]
keys=["split_image","ln_gelu_drop_fwd","ln_gelu_drop_bwd","gate_bwd_dz","gate_bwd_gemm","gate_bwd_dx","gate_bwd_dw","linear_bwd","linear_dw","gate_fwd","linear_fwd"]
print("$2: step %.3f ms | " % d["ms_per_step"] + " ".join("%s %.3f" % (x, k[x]) for x in keys if x in k))
PY
}
for rep in 1 2; do
  for V in actstride base; do
    if [ $V = base ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$R/tools/ab/$V.so; fi
    timeout 200 python bench.py $B > $OUT/${V}_$rep.json 2>/dev/null; show $OUT/${V}_$rep.json "$V $rep"
  done
done
unset MADELEINE_LIB
for M in none 0xFFFFFFFF 0x77777777 0x55555555 0x11111111; do
  if [ $M = none ]; then unset MADELEINE_DW_CUMASK; else export MADELEINE_DW_CUMASK=$M; fi
  MADELEINE_DW_STREAM=1 timeout 200 python bench.py $B > $OUT/dw_$M.json 2>$OUT/dw_$M.err; show $OUT/dw_$M.json "dw stream mask=$M" || tail -3 $OUT/dw_$M.err
done
