cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== p=0.1"; python $R/tools/exp_ln.py
echo "== p=0"; LN_P=0 python $R/tools/exp_ln.py
rm -rf /tmp/prof_ln
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d /tmp/prof_ln -- python $R/tools/exp_ln.py > /tmp/prof_ln.log 2>&1
python $R/tools/rocpd_summary.py /tmp/prof_ln/*/*.db 12 2>&1 | grep -v "^$" | cut -c1-200 | grep -i "ln_gelu\|drop_\|PMC\|kernel " | head -80
