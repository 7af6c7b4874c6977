#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_slab_axes.txt
: > $O
for rep in 1 2; do
  for mask in 0 4 7; do
    MVS_FFT_SLAB_AXES=$mask timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
    python - $mask <<'PY' | tee -a $O
import json, sys
r = json.load(open("gpurun_out/b.json"))
c = r["config"]
print("MVS_FFT_SLAB_AXES=%s ms_per_step %.2f register %.2f pairwise %.2f fuse %.2f" % (sys.argv[1], r["ms_per_step"], c.get("register_ms_per_step", float("nan")), c.get("pairwise_ms_per_step"), c.get("fuse_ms_per_step")))
PY
  done
done
