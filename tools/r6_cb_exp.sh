#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_fuse_gpu.py -x -q -k "content" 2>&1 | tail -3
python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
MVS_CB_GRID=2,4,4 MVS_CB_TILE=256,512,512 python tools/cb_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
python tools/cb_at_size.py 2>&1 | grep "options\|Error\|Skipped"
