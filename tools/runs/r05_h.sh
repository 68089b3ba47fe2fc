#!/bin/bash
# gate forward epilogue: folded activations + one select per pair + byte-field hash -- test suite, then step / kernel times (byte fields on / off, fp32 + bf16)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05h}; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extra-legs --no-bf16-leg"
show() {
python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1]); k=d["kernel_ms"]
print("$2: step %.3f ms | " % d["ms_per_step"] + " ".join("%s %.3f" % (x, k[x]) for x in ("gate_fwd","gate_bwd_dz","gate_bwd_gemm","linear_fwd","ln_gelu_drop_fwd") if x in k))
PY
}
for rep in 1 2; do
  for V in 16bit bytes; do
    if [ $V = bytes ]; then unset MADELEINE_DROP_16BIT; else export MADELEINE_DROP_16BIT=1; fi
    timeout 200 python bench.py $B > $OUT/f32_${V}_$rep.json 2>/dev/null; show $OUT/f32_${V}_$rep.json "f32 $V $rep"
    timeout 200 python bench.py $B --precision bfloat16 > $OUT/bf16_${V}_$rep.json 2>/dev/null; show $OUT/bf16_${V}_$rep.json "bf16 $V $rep"
  done
done
