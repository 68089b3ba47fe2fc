cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_l
rocprofv3 --kernel-trace --stats -d /tmp/prof_l -- python $R/tools/exp_linear_bf16.py > /tmp/prof_l.log 2>&1
python $R/tools/rocpd_summary.py /tmp/prof_l/*/*.db 14 2>&1 | grep -v "^$" | cut -c1-200 | head -40
