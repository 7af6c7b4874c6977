#!/bin/bash
# lanes with CU-masked streams: register() wall on the north-star mosaic
cd $GRAFT_REPO_ROOT
for cfg in "off 8" "xcd1 8" "xcd2 8" "xcd2 4" "xcd4 8" "spread2 8" "spread4 8" "xcd1 12" "off 8"; do
  set -- $cfg
  m=$1; lanes=$2
  if [ $m = off ]; then unset MVS_LANE_CU_MASK; else export MVS_LANE_CU_MASK=$m; fi
  echo "MVS_LANE_CU_MASK=$m $(timeout 300 python tools/sched_probe.py auto $lanes 8 2>&1 | tail -1)"
done
