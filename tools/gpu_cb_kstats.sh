#!/bin/bash
# kernel statistics of the content-based probe (tools/cb_probe.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cbks; rm -rf $O; mkdir -p $O
MVS_SERIAL=${SERIAL:-0} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python $R/tools/cb_probe.py > $O/probe.log 2>&1
tail -3 $O/probe.log
python $R/tools/kstats.py $(find $O/t -name "*kernel_stats.csv") 14 | grep -v "elementwise\|avg_pool"
python $R/tools/kgrid.py $(find $O/t -name "*kernel_trace.csv") "gauss" 24
find $O/t -name "*kernel_trace.csv" -delete; find $O/t -name "*.db" -delete
