#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per run) of the fuse launch with the streaming row kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/streampmc; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  MVS_STREAM=1 timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region|stream" --pmc $c --output-format csv -d $O/pmc_$c -- python $R/tools/fuse_probe.py 2 0 > $O/pmc_$c.log 2>&1
  echo "== stream $c"; grep -h "kernel ms" $O/pmc_$c.log | tail -1; python $R/tools/pmc_summary.py $(find $O/pmc_$c -name "*counter_collection.csv")
done > $O/summary.txt 2>&1
find $O -name "*.db" -delete
