#!/bin/bash
# the whole -m gpu suite + smoke + a bench line of the final tree (the line carries roofline.traffic when profiles/round5_fuse_traffic.json's
# digest equals the csrc digest)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r5_full_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r5_full_gpu.txt
timeout 600 python bench.py > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err
head -c 400 gpurun_out/r5_final_bench.json
