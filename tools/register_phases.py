"""Wall time of the phases of registration.register on the north-star mosaic (tiles resident)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import _lib, registration, mv_graph, param_resolution
from multiview_stitcher_amd import spatial_image_utils as si

dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
import gc; gc.collect(); gc.freeze()
T = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[name] = T.get(name, 0.0) + (time.perf_counter() - t0)
    setattr(mod, name, g)
wrap(mv_graph, "build_view_adjacency_graph"); wrap(mv_graph, "prune_view_adjacency_graph")
wrap(registration, "compute_pairwise_registrations"); wrap(param_resolution, "groupwise_resolution")
wrap(registration, "_prebin_views"); wrap(_lib, "synchronize")
for rep in range(6):
    T.clear()
    t0 = time.perf_counter()
    registration.register(sims, transform_key=key, new_transform_key="reg", device=0)
    tot = time.perf_counter() - t0
    if rep >= 3:
        print("register %.1f ms:" % (tot * 1e3), {k: round(v * 1e3, 1) for k, v in T.items()}, "other %.1f" % ((tot - sum(T.values())) * 1e3))
