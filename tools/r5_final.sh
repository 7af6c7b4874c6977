#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5final; mkdir -p $O
timeout 900 python -m pytest tests/test_register_fuse_gpu.py tests/test_sharding_gpu.py tests/test_abi_gpu.py tests/test_reg_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
bash tools/profile_round5.sh > $O/profile.log 2>&1; tail -2 $O/profile.log | head -c 300
