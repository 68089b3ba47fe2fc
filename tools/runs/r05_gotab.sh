#!/bin/bash
# GOT: A/B of sweep variants (same box) + per-kernel breakdown (rocprofv3) at n = 192 and n = 256
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05gotab}; mkdir -p $OUT
cd $R
L=$R/madeleine_amd/csrc/libmadeleine_amd.so
GOT_GEOMS='[[32,128],[25,188],[32,192],[32,256],[128,256]]' timeout 600 python tools/got_ab.py $R/tools/ab/r4base.so $L $R/tools/ab/hlr0.so $R/tools/ab/hlr7.so $R/tools/ab/rg4.so $R/tools/ab/rg1.so > $OUT/got_ab.txt 2>&1
cat $OUT/got_ab.txt | grep -v Warn
cd /tmp && export TMPDIR=/tmp
for g in "32,192" "128,256"; do
  rm -rf /tmp/pg; GOT_GEOMS="[[$g]]" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pg -- python $R/tools/bench_got.py > /tmp/pg.log 2>&1
  python $R/tools/rocpd_summary.py /tmp/pg/*/*.db 30 2>&1 | cut -c 1-170 > $OUT/got_kernels_${g/,/_}.txt
  cat $OUT/got_kernels_${g/,/_}.txt | cut -c 30-150
done
