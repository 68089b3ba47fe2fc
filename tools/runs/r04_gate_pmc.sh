#!/bin/bash
# SQ counters of the gate forward kernels (bf16 128 / 256 tile, split engine): what is busy while the kernel runs?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rm -rf /tmp/prof_$name; rocprofv3 "$@" > /tmp/prof_$name.log 2>&1
  { echo "# r04 $name: rocprofv3 $*" | sed "s#$R/##g"; python $R/tools/rocpd_summary.py /tmp/prof_$name/*/*.db 10 | grep -v "Cat\|elementwise\|copyBuffer\|fillBuffer\|convert\|absmax\|scale_kernel\|finalize\|gate_w"; } > $OUT/$name.txt 2>&1; tail -2 /tmp/prof_$name.log | head -1; rm -rf /tmp/prof_$name; }
for v in "--tile 256" "--tile 128"; do
  tag=$(echo $v | tr -d ' -')
  run gate_${tag}_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/prof_gate_${tag}_sq1 -- python $R/tools/prof_gate_bf16.py $v
  run gate_${tag}_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/prof_gate_${tag}_sq2 -- python $R/tools/prof_gate_bf16.py $v
  run gate_${tag}_sq3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU --kernel-trace -d /tmp/prof_gate_${tag}_sq3 -- python $R/tools/prof_gate_bf16.py $v
done
cat $OUT/*.txt | cut -c1-170
