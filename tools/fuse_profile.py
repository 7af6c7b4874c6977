"""cProfile of fusion.fuse on the north-star mosaic (host time around the fuse launch)."""
import sys, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import fusion, _lib
from multiview_stitcher_amd import spatial_image_utils as si
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
import gc; gc.collect(); gc.freeze()
for _ in range(3):
    out = fusion.fuse(sims, transform_key=key, output_on_backend=True, device=0); _lib.synchronize(0)
pr = cProfile.Profile(); pr.enable()
out = fusion.fuse(sims, transform_key=key, output_on_backend=True, device=0)
pr.disable()
_lib.synchronize(0)
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
