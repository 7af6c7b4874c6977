"""Register ONE pair of north-star sized tiles (for rocprofv3 --kernel-trace): python tools/reg_probe.py [reps] [grid z,y,x: the axis along which the two tiles are neighbours]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
_lib.init(0)
grid, tile = np.array([int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,1,2").split(",")]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=7)
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
for r in range(reps + 1):
    t0 = time.perf_counter()
    registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg", device=0,
                          pre_registration_pruning_method="keep_axis_aligned")
    torch.cuda.synchronize()
    print("register pair: %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
if os.environ.get("MVS_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for r in range(10):
        registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg", device=0,
                              pre_registration_pruning_method="keep_axis_aligned")
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(40)
