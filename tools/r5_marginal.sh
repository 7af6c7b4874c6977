#!/bin/bash
# marginal cost of every kernel inside the 8-lane pair loop: profiling build (-DMVS_PROFILING_ABLATIONS), MVS_DUP_KERNELS=<tag> launches
# that kernel twice (idempotent launches), the difference in the pairwise wall / (144 x launches per pair) = what one launch costs the loop
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_marginal.txt
: > $O
cp multiview-stitcher_amd/libmvs_hip.so /tmp/base.so
cp tools/variants/libmvs_hip_prof.so multiview-stitcher_amd/libmvs_hip.so
run() {  # tag, slab axes
  MVS_DUP_KERNELS=$1 MVS_FFT_SLAB_AXES=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
  python - "$1" "$2" <<'PY' | tee -a $O
import json, sys
r = json.load(open("gpurun_out/b.json"))
c = r["config"]
print("dup %-12s slab_axes %s ms_per_step %.2f register %.2f pairwise %.2f fuse %.2f" % (sys.argv[1], sys.argv[2], r["ms_per_step"], c.get("register_ms_per_step", float("nan")), c.get("pairwise_ms_per_step"), c.get("fuse_ms_per_step")))
PY
}
for rep in 1 2; do
  for tag in none ssim_fused ssim_fixed crop rescale hist_count hist_corr shift updft; do run $tag 0; done
  for tag in none slab long_xp; do run $tag 7; done
done
cp /tmp/base.so multiview-stitcher_amd/libmvs_hip.so
