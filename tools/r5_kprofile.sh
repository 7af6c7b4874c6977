#!/bin/bash
# round 5: kernel statistics of the bench command with the native pair loop (8 lanes = default, and 1 lane = isolated durations),
# GPU busy share / mean concurrency inside the registration window
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5prof; rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- $BENCH > $O/bench.log 2>&1
python $R/tools/kstats.py $(find $O/bench -name "*kernel_stats.csv") 60 > $O/bench_kstats.txt
python $R/tools/kbusy.py $(find $O/bench -name "*kernel_trace.csv") > $O/bench_busy.txt 2>&1
python $R/tools/fuse_window.py $(find $O/bench -name "*kernel_trace.csv") > $O/fuse_launch_windows.csv 2>&1
cp $(find $O/bench -name "*kernel_stats.csv") $O/bench_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench1 -- $BENCH --reg-threads 1 > $O/bench1.log 2>&1
python $R/tools/kstats.py $(find $O/bench1 -name "*kernel_stats.csv") 60 > $O/bench1_kstats.txt
cp $(find $O/bench1 -name "*kernel_stats.csv") $O/bench1_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -1 $O/bench.log | head -c 600; echo; cat $O/bench_busy.txt; grep -v "elementwise\|avg_pool\|distribution" $O/bench_kstats.txt | head -32
