#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ax in 4 0 7; do
  echo "axes $ax"
  MVS_FFT_SLAB_AXES=$ax timeout 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/b.err | head -c 300; echo " rc=$?"
done
