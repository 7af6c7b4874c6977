#!/bin/bash
# native pair loop: lanes sweep, wait modes, against the per-pair interpreter threads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
for mode in auto spin yield blocking; do
  for lanes in 16 8; do
    timeout 300 python tools/sched_probe.py $mode $lanes 2>&1 | tail -1
  done
done
for lanes in 4 6 12; do timeout 300 python tools/sched_probe.py auto $lanes 2>&1 | tail -1; done
MVS_NO_BATCH=1 timeout 300 python tools/sched_probe.py auto 16 2>&1 | tail -1
