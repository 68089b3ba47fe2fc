cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pg -- python $GRAFT_REPO_ROOT/tools/bench_got.py > /tmp/pg.log 2>&1
tail -6 /tmp/pg.log
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pg/*/*.db 30 2>&1 | cut -c 1-150
