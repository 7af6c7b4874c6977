#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "MVS_CBF_DBG=0" "MVS_CBF_DBG=2" "MVS_CBF_DBG=4" "MVS_CBF_DBG=1"; do
  echo "== $cfg"; env $cfg python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
done
