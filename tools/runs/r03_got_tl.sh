#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt; GOT_ONLY_MULTI=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/pt -- python $GRAFT_REPO_ROOT/tools/exp_got_overlap.py > /tmp/pt.log 2>&1
tail -3 /tmp/pt.log
python $GRAFT_REPO_ROOT/tools/got_timeline.py /tmp/pt/*/*.db > $GRAFT_REPO_ROOT/gpurun_out/got_timeline.txt 2>&1
wc -l $GRAFT_REPO_ROOT/gpurun_out/got_timeline.txt; tail -2 $GRAFT_REPO_ROOT/gpurun_out/got_timeline.txt
