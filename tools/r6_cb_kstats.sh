#!/bin/bash
# round 6: kernel statistics of the content-based probe (fast path; MVS_CB_EXACT=1 for the bit-faithful passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6cbk; rm -rf $O; mkdir -p $O
for m in ${MODES:-0}; do
  MVS_CB_EXACT=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$m -- python $R/tools/cb_probe.py > $O/m$m.log 2>&1
  echo "== cb_exact=$m" | tee -a $O/kstats.txt; python $R/tools/kstats.py $(find $O/m$m -name "*kernel_stats.csv") 24 | grep -v "elementwise\|avg_pool\|distribution" | head -18 | tee -a $O/kstats.txt
  python $R/tools/kgrid.py $(find $O/m$m -name "*kernel_trace.csv") "cb_line|gauss" 24 | tee -a $O/kstats.txt
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
