#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5cb; mkdir -p $O
timeout 900 python -m pytest tests/test_fuse_gpu.py -x -q -m gpu -k "content_based" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
for m in 1 0 1 0; do echo "cb_mask_closed_form=$m: $(MVS_CB_MASK=$m python tools/cb_probe.py 2>&1 | tail -2 | tr '\n' ' ')"; done
