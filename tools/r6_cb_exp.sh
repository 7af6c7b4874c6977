#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_fuse_gpu.py -x -q -k "content" 2>&1 | tail -3
for cfg in "MVS_CBF_DBG=0" "MVS_CBF_DBG=1"; do
  echo "== $cfg"; env $cfg python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
done
