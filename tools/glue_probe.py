"""Host-side cost of one pairwise registration (the part the GIL serialises across the worker threads):
python tools/glue_probe.py  -> cProfile of register_pair_of_msims on the main thread, 2x2x2 tiles of 512^3."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si

dev = torch.device("cuda", 0)
_lib.init(0)
grid, tile = np.array([2, 2, 2]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=7)
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
edges = [(0, 1), (0, 2), (0, 4), (1, 3), (1, 5), (2, 3), (2, 6), (3, 7), (4, 5), (4, 6), (5, 7), (6, 7)]
kw = dict(transform_key=si.DEFAULT_TRANSFORM_KEY, registration_binning=None, overlap_tolerance=0.0)
cache = registration._BinCache()
for i, j in edges:
    registration.register_pair_of_msims(sims[i], sims[j], device=0, _bin_cache=cache, **kw)
t0 = time.perf_counter()
for r in range(5):
    for i, j in edges:
        registration.register_pair_of_msims(sims[i], sims[j], device=0, _bin_cache=cache, **kw)
print("per pair %.3f ms wall" % ((time.perf_counter() - t0) / 60 * 1e3))
pr = cProfile.Profile()
pr.enable()
for r in range(5):
    for i, j in edges:
        registration.register_pair_of_msims(sims[i], sims[j], device=0, _bin_cache=cache, **kw)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(22)
