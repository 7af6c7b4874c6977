"""Where the first register() + fuse() of a process spend their time (config.step_cold_ms of the bench line).
python tools/cold_probe.py            (under `rocprofv3 --hip-trace --stats` for the runtime calls behind it)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multiview_stitcher_amd import _lib, fusion, registration
from multiview_stitcher_amd import spatial_image_utils as si

t0 = time.perf_counter(); _lib.init(0); print("mvs_init lane 0 %.1f ms" % ((time.perf_counter() - t0) * 1e3))
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
tiles, jit, org = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=1000)
sims = bench.build_sims(tiles, org, 0)
torch.cuda.synchronize()
if os.environ.get("MVS_PREINIT"):
    t0 = time.perf_counter()
    for lane in range(1, 8):
        _lib.init(lane << 8)
    print("mvs_init lanes 1-7 %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(3):
    t0 = time.perf_counter()
    registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="r", device=0)
    t1 = time.perf_counter()
    out = fusion.fuse(sims, transform_key="r", output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
    _lib.synchronize(0)
    t2 = time.perf_counter()
    print("call %d: register %.1f ms, fuse %.1f ms; pool misses lane0 %d (%.1f GB)" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, _lib.get_counter("pool_misses", 0, reset=True), _lib.get_counter("pool_miss_bytes", 0, reset=True) / 1e9),
          " lanes:", [int(_lib.get_counter("pool_misses", l << 8, reset=True)) for l in range(1, 8)])
    del out
