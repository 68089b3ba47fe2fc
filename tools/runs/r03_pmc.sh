#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rm -rf /tmp/prof_$name; rocprofv3 "$@" > /tmp/prof_$name.log 2>&1
  { echo "# r03e $name: rocprofv3 $*" | sed "s#$R/##g"; python $R/tools/rocpd_summary.py /tmp/prof_$name/*/*.db 60; } > $OUT/r03e_$name.txt 2>&1; rm -rf /tmp/prof_$name; }
run split_pmc_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d /tmp/prof_split_pmc_sq -- python $R/tools/prof_split.py --iters 1
run split_pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d /tmp/prof_split_pmc_inst -- python $R/tools/prof_split.py --iters 1
run split_pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE WRITE_SIZE --kernel-trace -d /tmp/prof_split_pmc_tcc -- python $R/tools/prof_split.py --iters 1
grep -A200 "# PMC" $OUT/r03e_split_pmc_sq.txt | grep "sp_" | cut -c 1-160
grep -A200 "# PMC" $OUT/r03e_split_pmc_inst.txt | grep "sp_" | cut -c 1-160
grep -A200 "# PMC" $OUT/r03e_split_pmc_tcc.txt | grep "sp_" | cut -c 1-160
