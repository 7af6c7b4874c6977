"""BASELINE config C2: 3x3 grid of 2048^2 float32 tiles, 20 % overlap, register + cosine-blend fuse (2D)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiview_stitcher_amd import _lib, fusion, registration, sample_data, param_utils
from multiview_stitcher_amd import spatial_image_utils as si
from multiview_stitcher_amd.device import DeviceArray
_lib.init(0)
sims, jit, _ = sample_data.generate_tiled_dataset(ndim=2, tile_shape=(2048, 2048), tiles=(3, 3), overlap=(410, 410), dtype=np.float32, seed=0)
sims = [s.isel({"c": 0, "t": 0}) if "c" in s.dims else s for s in sims]
sims = [s.copy(data=DeviceArray.from_host(np.ascontiguousarray(s.data), 0)) for s in sims]
for rep in range(3):
    t0 = time.perf_counter()
    registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg", device=0, pre_registration_pruning_method="keep_axis_aligned")
    t1 = time.perf_counter()
    out = fusion.fuse(sims, transform_key="reg", output_chunksize={d: 2048 for d in "yx"}, output_on_backend=True, device=0)
    _lib.synchronize(0)
    t2 = time.perf_counter()
    rec = np.array([param_utils.select_time(si.get_affine_from_sim(s, "reg"), 0)[:2, 2] for s in sims])
    print("C2 %s: register %.1f ms, fuse %.1f ms, %.0f Mvoxels/s; max |recovered - jitter| = %.3g px" % (
        out.shape, (t1 - t0) * 1e3, (t2 - t1) * 1e3, np.prod(out.shape) / (t2 - t0) / 1e6, np.abs((rec - rec[0]) - (jit - jit[0])).max()), flush=True)   # relative to tile 0: the resolver fixes its own reference view
if os.environ.get("MVS_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        out = fusion.fuse(sims, transform_key="reg", output_chunksize={d: 2048 for d in "yx"}, output_on_backend=True, device=0)
    _lib.synchronize(0)
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
