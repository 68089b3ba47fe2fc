#!/bin/bash
# rocprofv3 kernel trace of one rank of config 4 (emulated global batch): concurrency of the four stains' GOT chains inside a step
TAG=${1:-r04e}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_txt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c4
BENCH_NO_TIMER=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_c4 -- python $R/tools/debug/prof_c4_leg.py > /tmp/prof_c4.log 2>&1
tail -3 /tmp/prof_c4.log
for db in /tmp/prof_c4/*/*.db; do
  { echo "# $TAG c4 rank emulation, GOT kernels of one step (window: FusedAdam occurrence 7 -> 8)"; python $R/tools/rocpd_phase.py $db got_ FusedAdam 7 8; } > $OUT/${TAG}_c4_got_concurrency.txt 2>&1
  python $R/tools/rocpd_timeline.py $db got_prep 5 700 8 > $OUT/${TAG}_c4_got_timeline.txt 2>&1
done
cat $OUT/${TAG}_c4_got_concurrency.txt
