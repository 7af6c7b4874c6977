#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5launch; mkdir -p $O
{ echo "== empty kernels, sync every 4"; ./tools/ubench/launch_rate 4000 4 0
  echo "== empty kernels, sync every 1"; ./tools/ubench/launch_rate 3000 1 0
  echo "== empty kernels, sync every 32"; ./tools/ubench/launch_rate 4000 32 0
  echo "== ~20 us kernels (256 blocks), sync every 4"; ./tools/ubench/launch_rate 2000 4 3000
} > $O/launch_rate.txt 2>&1
cat $O/launch_rate.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5gaps; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $O/bench.log 2>&1
python $R/tools/lane_gaps.py $(find $O/t -name "*kernel_trace.csv") > $O/lane_gaps.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/lane_gaps.txt
