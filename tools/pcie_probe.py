"""PCIe-inclusive rate of the north-star workload: tiles start in pinned host memory, are uploaded
(one H2D per tile, shared by register and fuse), registered, fused, and the mosaic is copied back to pinned host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multiview_stitcher_amd import _lib, fusion, registration
from multiview_stitcher_amd import spatial_image_utils as si

dev = torch.device("cuda", 0)
_lib.init(0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=1000)
host_tiles = [t.view(torch.int16).cpu().pin_memory() for t in tiles]
del tiles
torch.cuda.synchronize()
out_host = None
for rep in range(3):
    t0 = time.perf_counter()
    dtiles = [h.to(dev, non_blocking=True).view(torch.uint16) for h in host_tiles]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sims = bench.build_sims(dtiles, origins, 0)
    registration.register(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, new_transform_key="reg", device=0,
                          pre_registration_pruning_method="keep_axis_aligned")
    t2 = time.perf_counter()
    fused = fusion.fuse(sims, transform_key="reg", output_chunksize={d: 1 << 30 for d in "zyx"}, output_on_backend=True, device=0)
    _lib.synchronize(0)
    t3 = time.perf_counter()
    ft = torch.as_tensor(type("V", (), {"__cuda_array_interface__": dict(fused.data.__cuda_array_interface__, typestr="<i2")})(), device="cuda")
    if out_host is None:
        out_host = torch.empty(ft.shape, dtype=torch.int16).pin_memory()
    out_host.copy_(ft, non_blocking=True)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    vox = float(np.prod(ft.shape))
    print("rep %d: H2D %.0f ms (%.1f GB/s), register %.0f ms, fuse %.0f ms, D2H %.0f ms (%.1f GB/s); total %.0f ms = %.0f Mvoxels/s PCIe-inclusive"
          % (rep, (t1 - t0) * 1e3, 64 * 2 ** 28 / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, vox * 2 / (t4 - t3) / 1e9,
             (t4 - t0) * 1e3, vox / (t4 - t0) / 1e6), flush=True)
    del dtiles, sims, fused, ft
