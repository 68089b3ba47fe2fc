#!/bin/bash
# PMC passes over the split NT product alone (tools/prof_split_nt.py): in-flight levels (= latency x rate) of VMEM / LDS instructions
TAG=${1:-r04g}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/pmc_nt
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_nt -- python $R/tools/prof_split_nt.py 2 > /tmp/pmc_nt.log 2>&1
  { echo "# $TAG split_nt_pmc$i: rocprofv3 --pmc $C --kernel-trace -- python tools/prof_split_nt.py 2 (K 2048 -> N 512 and K 512 -> N 2048, T = 262144, 2 launches each)"; python $R/tools/rocpd_summary.py /tmp/pmc_nt/*/*.db 6; tail -2 /tmp/pmc_nt.log | cut -c1-200; } > $OUT/${TAG}_split_nt_pmc$i.txt 2>&1
  grep -E "sp_nt_kernel" $OUT/${TAG}_split_nt_pmc$i.txt | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}'
done
