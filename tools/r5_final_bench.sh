#!/bin/bash
# bench line of the final tree (roofline.traffic needs profiles/round5_fuse_traffic.json's digest == csrc digest) + launch tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err
tail -c 600 gpurun_out/r5_final_bench.json
timeout 900 python -m pytest tests/test_bench_launch_gpu.py -x -q -m gpu 2>&1 | tail -4
