#!/bin/bash
# what of the aligned geometry's speed is the output pitch, what the store alignment, what the load alignment (same kernels)
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" MVS_SERIAL=${SER:-0} python tools/fuse_probe.py 6 $G 4,4,4 512,512,512 $OV 2>&1 | grep "kernel ms\|class" | cut -c1-200; }
for SER in 0 1; do
G=2 OV=102; run "bench geometry (ov 102, jitter), pitch 3496" A=1
G=2 OV=102; run "bench geometry, rows padded to 64 voxels (pitch 3584)" MVS_PAD_X=64
G=0 OV=128; run "ov 128 exact grid: everything 128-B aligned" A=1
G=0 OV=128; run "ov 128, output origin 3 voxels to the left, rows padded: stores misaligned by 6 B, loads aligned" MVS_PAD_X=64 MVS_SHIFT_X=3
G=0 OV=128; run "ov 128, origin 3 voxels to the left, NOT padded (pitch 3334): stores misaligned + pitch" MVS_SHIFT_X=3
done
