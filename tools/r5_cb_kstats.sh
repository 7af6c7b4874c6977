#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5cbk; rm -rf $O; mkdir -p $O
for m in 1 0; do
  MVS_CB_MASK=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$m -- python $R/tools/cb_probe.py > $O/m$m.log 2>&1
  echo "== cb_mask_closed_form=$m"; python $R/tools/kstats.py $(find $O/m$m -name "*kernel_stats.csv") 22 | grep -v "elementwise\|avg_pool\|distribution" | head -8
  python $R/tools/kgrid.py $(find $O/m$m -name "*kernel_trace.csv") "gauss|bbox" 14
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
