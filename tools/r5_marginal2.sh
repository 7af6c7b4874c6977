#!/bin/bash
# marginal cost of the small kernels of a pair (see tools/r5_marginal.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5_marginal2.txt
: > $O
cp multiview-stitcher_amd/libmvs_hip.so /tmp/base.so
cp tools/variants/libmvs_hip_prof.so multiview-stitcher_amd/libmvs_hip.so
run() {
  MVS_DUP_KERNELS=$1 timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
  python - "$1" <<'PY' | tee -a $O
import json, sys
r = json.load(open("gpurun_out/b.json")); c = r["config"]
print("dup %-28s ms_per_step %.2f register %.2f pairwise %.2f fuse %.2f" % (sys.argv[1], r["ms_per_step"], c.get("register_ms_per_step"), c.get("pairwise_ms_per_step"), c.get("fuse_ms_per_step")))
PY
}
for rep in 1 2 3; do
  for tag in none finish updft_mid hist_fold rank_table finish,updft_mid,hist_fold,rank_table; do run $tag; done
done
cp /tmp/base.so multiview-stitcher_amd/libmvs_hip.so
