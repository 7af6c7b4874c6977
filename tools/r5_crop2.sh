#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_register_fuse_gpu.py -x -q -m gpu -k "batched or binning" 2>&1 | tail -3
bash tools/r5_crop_orient.sh
for r in 1 0 1 0; do echo "MVS_RAW_CROPS=$r $(MVS_RAW_CROPS=$r timeout 300 python tools/sched_probe.py auto 8 8 2>&1 | tail -1)"; done
