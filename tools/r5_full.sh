#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5full; mkdir -p $O; rm -f $O/at_size.jsonl
MVS_AT_SIZE_STATS=$GRAFT_REPO_ROOT/$O/at_size.jsonl timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
