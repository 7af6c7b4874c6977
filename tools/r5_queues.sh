#!/bin/bash
# native pair loop: hardware queues x lanes
cd $GRAFT_REPO_ROOT
for q in 4 8 16; do
  for lanes in 8 12; do
    echo "GPU_MAX_HW_QUEUES=$q $(GPU_MAX_HW_QUEUES=$q timeout 300 python tools/sched_probe.py auto $lanes 8 2>&1 | tail -1)"
  done
done
echo "GPU_MAX_HW_QUEUES=2 $(GPU_MAX_HW_QUEUES=2 timeout 300 python tools/sched_probe.py auto 8 8 2>&1 | tail -1)"
echo "GPU_MAX_HW_QUEUES=8 lanes 6 $(GPU_MAX_HW_QUEUES=8 timeout 300 python tools/sched_probe.py auto 6 8 2>&1 | tail -1)"
