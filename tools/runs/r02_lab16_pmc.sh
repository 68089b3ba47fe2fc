cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_lab
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/prof_lab -- $R/tools/micro/build/gemm_lab_bf16 2 > /tmp/prof_lab.log 2>&1
python $R/tools/rocpd_summary.py /tmp/prof_lab/*/*.db 40 2>&1 | grep -i "tn128\|tn256\|nt256\|nt128PK\|PMC" | cut -c1-180
