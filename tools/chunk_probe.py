"""North-star fuse through the chunked workflow (the reference's way: one fuse_np per output chunk)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multiview_stitcher_amd import _lib, fusion
from multiview_stitcher_amd import spatial_image_utils as si
dev = torch.device("cuda", 0); _lib.init(0)
grid, tile = np.array([4, 4, 4]), np.array([512] * 3)
tiles, jit, org = bench.make_mosaic_on_device(torch, dev, grid, tile, np.round(tile * 0.2).astype(int), seed=1000)
sims = bench.build_sims(tiles, org, 0)
torch.cuda.synchronize()   # the tiles are produced on torch's stream, the library runs on its own
for merge in (True, False):
    for cs in (1 << 30, 1024, 512, 256):
        for rep in range(2):
            t0 = time.perf_counter()
            out = fusion.fuse(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, output_chunksize={d: cs for d in "zyx"}, output_on_backend=True, device=0,
                              merge_chunks=merge)
            _lib.synchronize(0)
            dt = time.perf_counter() - t0
        print("merge_chunks=%s chunksize %d: %.1f ms (%.0f Mvoxels/s)" % (merge, cs, dt * 1e3, np.prod(out.shape) / dt / 1e6), flush=True)
