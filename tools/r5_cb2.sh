#!/bin/bash
cd $GRAFT_REPO_ROOT
MVS_CB_COUNT=1 MVS_CB_DEBUG=1 python tools/cb_probe.py 2>&1 | grep "cb mask" | head -10
