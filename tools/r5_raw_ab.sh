#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for r in 1 0; do echo "MVS_RAW_CROPS=$r $(MVS_RAW_CROPS=$r timeout 300 python tools/sched_probe.py auto 8 8 2>&1 | tail -1)"; done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5raw; rm -rf $O; mkdir -p $O
for r in 1 0; do
  MVS_RAW_CROPS=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$r -- python $R/tools/sched_probe.py auto 1 3 > $O/r$r.log 2>&1
  echo "== raw $r (1 lane)"; python $R/tools/kstats.py $(find $O/r$r -name "*kernel_stats.csv") 40 | grep -i "crop\|bin_mean\|rescale"
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
