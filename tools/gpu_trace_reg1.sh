#!/bin/bash
# kernel trace of register() on the north-star grid with ONE lane: isolated kernel durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace; mkdir -p $O; rm -rf $O/p1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p1 -- python $R/tools/host_cpu_probe.py 1 2 > $O/p1.log 2>&1
python $R/tools/kbusy2.py $(find $O/p1 -name "*kernel_trace.csv") | tee $O/busy1.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
