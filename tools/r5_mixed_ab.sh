#!/bin/bash
# alternating A/B of the fuse launch time: five class launches (0) / mixed list (1)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for m in 0 1; do
    echo "MVS_FUSE_MIXED=$m jittered: $(MVS_FUSE_MIXED=$m python tools/fuse_probe.py 8 2 2>&1 | grep 'kernel ms' | tail -1)"
  done
done
for m in 0 1; do echo "MVS_FUSE_MIXED=$m exact grid: $(MVS_FUSE_MIXED=$m python tools/fuse_probe.py 8 0 2>&1 | grep 'kernel ms' | tail -1)"; done
for m in 0 1; do echo "MVS_FUSE_MIXED=$m fractional: $(MVS_FUSE_MIXED=$m python tools/fuse_probe.py 6 1 2>&1 | grep 'kernel ms' | tail -1)"; done
