#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "" "MVS_CBF_DBG=1" "MVS_CBF_XT=8" "MVS_CBF_XT=16" "MVS_CBF_XT=8 MVS_CBF_DBG=1"; do
  echo "== $cfg"; env $cfg python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
done
