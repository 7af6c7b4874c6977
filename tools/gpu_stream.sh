#!/bin/bash
# A/B of the streaming row kernel against the region kernels (correctness, then time)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_fuse_gpu.py -x -q -k "stream" 2>&1 | tail -5
MVS_PLAN_STATS=1 MVS_STREAM=1 MVS_COMPARE=1 timeout 300 python tools/fuse_probe.py 2 2 2,3,3 256,384,640 2>&1 | grep -v "unit bricks\|regions" | tail -4
MVS_STREAM=1 MVS_COMPARE=1 timeout 300 python tools/fuse_probe.py 2 0 2,2,2 300,300,700 2>&1 | tail -2
for s in 0 1; do
  echo "== stream=$s exact grid"; MVS_STREAM=$s timeout 300 python tools/fuse_probe.py 6 0 2>&1 | tail -1
  echo "== stream=$s jitter";     MVS_STREAM=$s timeout 300 python tools/fuse_probe.py 6 2 2>&1 | tail -1
done
for a in 1 2 4; do
  echo "== stream=1 exact grid ablate=$a"; MVS_ABLATE=$a MVS_STREAM=1 timeout 300 python tools/fuse_probe.py 5 0 2>&1 | tail -1
done
} > gpurun_out/stream.log 2>&1
