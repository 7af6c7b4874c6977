#!/bin/bash
# the copy / one-view / two-view classes of the fuse launch as ONE launch over a space-ordered list
# (option "fuse_mixed", MVS_FUSE_MIXED=1) against the five class launches: launch time (HIP events), HBM traffic (FETCH_SIZE /
# WRITE_SIZE, one counter per run), and bit equality of the mosaic.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5mixed; rm -rf $O; mkdir -p $O
for m in 0 1; do
  export MVS_FUSE_MIXED=$m
  echo "== MVS_FUSE_MIXED=$m: launch ms (jittered geometry = the bench's, 6 launches)"; python $R/tools/fuse_probe.py 6 2 2>&1 | grep "kernel ms" | tail -1
  echo "== MVS_FUSE_MIXED=$m: exact grid"; python $R/tools/fuse_probe.py 6 0 2>&1 | grep "kernel ms" | tail -1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-include-regex "fuse|copy_region" --pmc $c --output-format csv -d $O/pmc_${m}_$c -- python $R/tools/fuse_probe.py 2 2 > $O/pmc_${m}_$c.log 2>&1
    echo "== MVS_FUSE_MIXED=$m $c"; python $R/tools/pmc_summary.py $(find $O/pmc_${m}_$c -name "*counter_collection.csv")
  done
done
unset MVS_FUSE_MIXED
cd $R; python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import _lib, fusion, spatial_image_utils as si
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([256, 256, 256])
tiles, jit, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=3)
sims = bench.build_sims(tiles, origins, 0)
rng = np.random.default_rng(0)
for s_, j in zip(sims, jit):
    p = np.eye(4); p[:3, 3] = j; si.set_sim_affine(s_, p, "reg")
outs = []
for m in (0, 1):
    _lib.set_option("fuse_mixed", m)
    fusion._REPLAY_MEMO.clear()
    outs.append(np.asarray(fusion.fuse(sims, transform_key="reg", output_on_backend=True, device=0).data))
_lib.set_option("fuse_mixed", 0)
print("mixed == class launches:", bool(np.array_equal(outs[0], outs[1])), outs[0].shape)
PY
