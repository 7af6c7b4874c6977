"""cProfile of the main thread of registration.register on the north-star mosaic (what the host does outside the pair workers)."""
import sys, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import registration
from multiview_stitcher_amd import spatial_image_utils as si
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
import gc; gc.collect(); gc.freeze()
for _ in range(3):
    registration.register(sims, transform_key=key, new_transform_key="reg", device=0)
pr = cProfile.Profile(); pr.enable()
registration.register(sims, transform_key=key, new_transform_key="reg", device=0)
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(32)
