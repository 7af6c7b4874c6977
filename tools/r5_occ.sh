#!/bin/bash
# occupancy / granularity knobs of the SSIM walk inside the 8-lane pair loop (one box, alternating)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_occ.txt
: > $O
cp multiview-stitcher_amd/libmvs_hip.so /tmp/base.so
run() {  # label
  MVS_FFT_SLAB_AXES=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
  python - "$1" <<'PY' | tee -a $O
import json, sys
r = json.load(open("gpurun_out/b.json"))
c = r["config"]
print("%-28s ms_per_step %.2f register %.2f pairwise %.2f fuse %.2f" % (sys.argv[1], r["ms_per_step"], c.get("register_ms_per_step", float("nan")), c.get("pairwise_ms_per_step"), c.get("fuse_ms_per_step")))
PY
}
for rep in 1 2; do
  cp /tmp/base.so multiview-stitcher_amd/libmvs_hip.so; run "wpe(3,4) items 320"
  MVS_SSIM_PRUNE_ITEMS=160 run "wpe(3,4) items 160"
  MVS_SSIM_PRUNE_ITEMS=640 run "wpe(3,4) items 640"
  cp tools/variants/libmvs_hip_wpe44.so multiview-stitcher_amd/libmvs_hip.so; run "wpe(4,4) items 320"
  cp tools/variants/libmvs_hip_wpe22.so multiview-stitcher_amd/libmvs_hip.so; run "wpe(2,2) items 320"
done
cp /tmp/base.so multiview-stitcher_amd/libmvs_hip.so
