"""Host-only timing of registration.register's serial section on the north-star geometry (no GPU): metadata-only tiles
(sharding.RemoteArray), a stub pairwise executor that returns consistent translations.  python tools/host_path_cpu.py [--profile]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import registration, param_utils
from multiview_stitcher_amd import spatial_image_utils as si

grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
step = tile - np.round(tile * 0.2).astype(int)
origins = [(np.array(idx) * step).astype(float) for idx in np.ndindex(*grid)]
sims = bench.build_sims([None] * 64, origins, 0, tile_shape=tile)
rng = np.random.default_rng(0)
jit = rng.integers(-3, 4, size=(64, 3)).astype(float)


def stub(msims, edges, kwargs):
    out = []
    for i, j in edges:
        lo = np.maximum(origins[i], origins[j])
        hi = np.minimum(origins[i], origins[j]) + 511
        out.append({"transform": param_utils.affine_from_translation(jit[j] - jit[i]), "quality": 0.95, "bbox": np.array([lo, hi])})
    return out


def run():
    return registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg", pairwise_executor=stub)


for flags in ((True, True), (False, False)):
    registration._native_graph[0], registration._native_resolution[0] = flags
    run()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); stub(None, [(0, 1)] * 144, None); ts_stub = time.perf_counter() - t0
    print("native graph/resolution", flags, "register() host ms %.2f (stub %.2f)" % (np.median(ts) * 1e3, ts_stub * 1e3))
registration._native_graph[0] = registration._native_resolution[0] = True
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): run()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
