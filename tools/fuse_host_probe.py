"""Host-side cost of fusion.fuse on the north-star mosaic (single chunk, 64 device-resident views): cProfile."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multiview_stitcher_amd import _lib, fusion
from multiview_stitcher_amd import spatial_image_utils as si

dev = torch.device("cuda", 0)
_lib.init(0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=7)
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
kw = dict(transform_key=si.DEFAULT_TRANSFORM_KEY, output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
for r in range(3):
    t0 = time.perf_counter()
    out = fusion.fuse(sims, **kw)
    t1 = time.perf_counter()
    _lib.synchronize(0)
    print("fuse call %.2f ms, +sync %.2f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    del out
pr = cProfile.Profile()
pr.enable()
for r in range(5):
    out = fusion.fuse(sims, **kw)
    _lib.synchronize(0)
    del out
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(30)
