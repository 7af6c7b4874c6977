#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
MVS_AT_SIZE_STATS=$GRAFT_REPO_ROOT/$O/crop_stats.jsonl timeout 900 python -m pytest tests/test_at_size_parity_gpu.py -x -q -m gpu -k "crop_length or c1_two" > $O/pytest.log 2>&1; echo "rc $?"; tail -30 $O/pytest.log; cat $O/crop_stats.jsonl
