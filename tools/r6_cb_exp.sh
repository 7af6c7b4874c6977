#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_fuse_gpu.py -x -q -k "content" 2>&1 | tail -3
for t in 1 0 1; do
  echo "== cb_taps_f64=$t"; MVS_CB_TAPS_F64=$t python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
done
python tools/cb_at_size.py 2>&1 | grep "options\|Error\|Skipped"
