"""merged launch blocks vs chunk-by-chunk with fractional offsets: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multiview_stitcher_amd import _lib, fusion
from multiview_stitcher_amd import spatial_image_utils as si
dev = torch.device("cuda", 0); _lib.init(0)
grid, tile = np.array([2, 2, 2]), np.array([256, 256, 256])
overlap = np.round(tile * 0.2).astype(int)
tiles, jit, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=7, max_jitter=0)
rng = np.random.default_rng(3)
origins = origins + rng.uniform(-1.5, 1.5, origins.shape)
sims = bench.build_sims(tiles, origins, 0)
for i, t in enumerate(tiles):
    t.view(torch.int16).add_(500 * (i % 8))
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
cs = {d: 128 for d in "zyx"}
a = fusion.fuse(sims, transform_key=key, output_chunksize=cs, output_on_backend=True, device=0)
b = fusion.fuse(sims, transform_key=key, output_chunksize=cs, output_on_backend=True, device=0, merge_chunks=False)
_lib.synchronize(0)
A = np.asarray(a.data).astype(np.int64); B = np.asarray(b.data).astype(np.int64)
d = A - B
print("shape", d.shape, "ndiff", (d != 0).sum(), "frac", (d != 0).mean(), "max", np.abs(d).max())
idx = np.argwhere(d != 0)
if len(idx):
    print("z range", idx[:, 0].min(), idx[:, 0].max(), "y", idx[:, 1].min(), idx[:, 1].max(), "x", idx[:, 2].min(), idx[:, 2].max())
    for ax in range(3):
        h = np.bincount(idx[:, ax] // 16, minlength=d.shape[ax] // 16 + 1)
        print("axis", ax, "hist/16:", h.tolist())
    print("examples", idx[:5].tolist(), [int(d[tuple(i)]) for i in idx[:5]])
for opt in ("force_generic",):
    _lib.set_option(opt, 1)
    a2 = fusion.fuse(sims, transform_key=key, output_chunksize=cs, output_on_backend=True, device=0)
    b2 = fusion.fuse(sims, transform_key=key, output_chunksize=cs, output_on_backend=True, device=0, merge_chunks=False)
    _lib.synchronize(0)
    d2 = np.asarray(a2.data).astype(np.int64) - np.asarray(b2.data).astype(np.int64)
    print(opt, "ndiff", (d2 != 0).sum(), "max", np.abs(d2).max())
    print("fast merged vs generic merged", (np.asarray(a2.data).astype(np.int64) != A).sum(), "fast chunk vs generic chunk", (np.asarray(b2.data).astype(np.int64) != B).sum())
    _lib.set_option(opt, 0)
