"""cProfile of the host side of one north-star step, sorted by CUMULATIVE time and restricted to the package's own functions:
which calls of register() / fuse() the interpreter time sits under."""
import cProfile, pstats, sys, io, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import _lib, fusion, registration
from multiview_stitcher_amd import spatial_image_utils as si

dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
reg = lambda: registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pre_registration_pruning_method="alternating_pattern")
fus = lambda: (fusion.fuse(sims, transform_key="reg", output_on_backend=True, device=0), _lib.synchronize(0))
reg(); fus(); reg(); fus()
import gc; gc.collect(); gc.freeze()
for name, fn in (("register", reg), ("fuse", fus)):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "ms", [round(t, 1) for t in ts])
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): fn()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("multiview", 40)
    for line in s.getvalue().splitlines():
        if "multiview" in line or "ncalls" in line:
            print(line[:60] + line[60:].split("multiview-stitcher_amd/")[-1][:70])
