"""cProfile of the host side of one north-star step (register + fuse), tiles resident: where the Python time goes."""
import cProfile, pstats, sys, io, time, os
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

def step():
    registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pre_registration_pruning_method="alternating_pattern")
    f = fusion.fuse(sims, transform_key="reg", output_on_backend=True, device=0)
    _lib.synchronize(0)
    return f

step(); step()
for name, fn in (("register", lambda: registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pre_registration_pruning_method="alternating_pattern")),
                 ("fuse", lambda: (fusion.fuse(sims, transform_key="reg", output_on_backend=True, device=0), _lib.synchronize(0)))):
    t0 = time.perf_counter(); fn(); print(name, "ms", (time.perf_counter() - t0) * 1e3)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): fn()
    pr.disable()
    st = pstats.Stats(pr)
    rows = sorted(((v[2], v[3], v[1], k) for k, v in st.stats.items()), reverse=True)[:45]
    print("   tottime_us  cumtime_us  calls   (per call of the profiled function, 10 calls averaged)")
    for tt, ct, nc, (fname, line, func) in rows:
        print(f"   {tt * 1e5:9.1f}  {ct * 1e5:9.1f}  {nc / 10:6.1f}   {os.path.basename(fname)}:{line}({func})")
