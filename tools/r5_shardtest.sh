#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sharding_gpu.py tests/test_bench_launch_gpu.py tests/test_register_fuse_gpu.py -x -q -m gpu 2>&1 | tail -8
