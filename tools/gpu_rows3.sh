#!/bin/bash
cd $GRAFT_REPO_ROOT
for ab in ${ABL:-0 4 8 12}; do echo "== ablate $ab"; MVS_ABLATE=$ab timeout 300 python tools/fuse_probe.py 3 0 2>&1 | tail -1; done
