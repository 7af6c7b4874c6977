#!/bin/bash
# A/B of two builds of the library on one box (alternating): tools/variants/libmvs_hip_before.so against the tree's libmvs_hip.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_ab_lib.txt
: > $O
cp multiview-stitcher_amd/libmvs_hip.so /tmp/new.so
run() {
  timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
  python - "$1" <<'PY' | tee -a $O
import json, sys
r = json.load(open("gpurun_out/b.json")); c = r["config"]
print("%-10s ms_per_step %.2f register %.2f pairwise %.2f fuse %.2f" % (sys.argv[1], r["ms_per_step"], c.get("register_ms_per_step"), c.get("pairwise_ms_per_step"), c.get("fuse_ms_per_step")))
PY
}
for rep in 1 2 3; do
  cp tools/variants/libmvs_hip_before.so multiview-stitcher_amd/libmvs_hip.so; run before
  cp /tmp/new.so multiview-stitcher_amd/libmvs_hip.so; run after
done
cp /tmp/new.so multiview-stitcher_amd/libmvs_hip.so
timeout 300 python -m pytest tests/test_reg_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
