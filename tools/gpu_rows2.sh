#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rows2; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/fuse_probe.py 3 ${1:-0} > $O/probe.log 2>&1
tail -2 $O/probe.log
python $R/tools/kstats.py $(find $O/kt -name "*kernel_stats.csv") 12
find $O/kt -name "*kernel_trace.csv" -delete; find $O/kt -name "*.db" -delete
