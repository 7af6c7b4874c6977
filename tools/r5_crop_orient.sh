#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5orient; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $R/tools/sched_probe.py auto 1 3 > $O/log.txt 2>&1
cd $R; python tools/crop_by_orientation.py $(find $O/t -name "*kernel_trace.csv"); python tools/kernels_by_orientation.py $(find $O/t -name "*kernel_trace.csv") | tee $O/kernels_by_orientation.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
