"""Content-based fusion (weights.content_based, C3-like): 2x2x2 grid of 256^3 u16 tiles, 256^3 output chunks (+ 22 px halo)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multiview_stitcher_amd import _lib, fusion
from multiview_stitcher_amd import spatial_image_utils as si
dev = torch.device("cuda", 0); _lib.init(0)
# MVS_CB_GRID / MVS_CB_TILE (z,y,x): another mosaic, e.g. BASELINE's C3 itself: MVS_CB_GRID=2,4,4 MVS_CB_TILE=256,512,512
grid = np.array([int(v) for v in os.environ.get("MVS_CB_GRID", "2,2,2").split(",")])
tile = np.array([int(v) for v in os.environ.get("MVS_CB_TILE", "256,256,256").split(",")])
overlap = np.round(tile * 0.2).astype(int)
tiles, jit, org = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=5, max_jitter=0)
sims = bench.build_sims(tiles, org, 0)
_lib.set_option("serial_classes", int(os.environ.get("MVS_SERIAL", "0")))      # 1: the views' filter chains one after the other (per-kernel timings)
_lib.set_option("cb_unpaired", int(os.environ.get("MVS_CB_UNPAIRED", "0")))
_lib.set_option("cb_nosplit", int(os.environ.get("MVS_CB_NOSPLIT", "0")))
_lib.set_option("cb_mask_closed_form", int(os.environ.get("MVS_CB_MASK", "0")))      # 1: masks that are boxes from tables (round 5; off by default)
_lib.set_option("cb_exact", int(os.environ.get("MVS_CB_EXACT", "0")))                # 1: the bit-faithful passes (rounds 1-5); default: the fast path (round 6)
if os.environ.get("MVS_CB_TAPS_F64"):
    _lib.set_option("cb_taps_f64", int(os.environ["MVS_CB_TAPS_F64"]))               # accumulators of the fast path: 1 float64 (default), 0 float32, 2 / 3 mixed
if os.environ.get("MVS_CB_COUNT"):
    _lib.set_option("cb_mask_count", 1)
torch.cuda.synchronize()   # the tiles are produced on torch's stream, the library runs on its own
if os.environ.get("MVS_CB_PROFILE"):      # where the interpreter spends a fuse() call
    import cProfile, pstats
    fusion.fuse(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, weights_func=fusion.content_based, output_chunksize={d: 256 for d in "zyx"}, output_on_backend=True, device=0)
    pr = cProfile.Profile(); pr.enable()
    fusion.fuse(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, weights_func=fusion.content_based, output_chunksize={d: 256 for d in "zyx"}, output_on_backend=True, device=0)
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
for rep in range(5):
    t0 = time.perf_counter()
    out = fusion.fuse(sims, transform_key=si.DEFAULT_TRANSFORM_KEY, weights_func=fusion.content_based, output_chunksize={d: 256 for d in "zyx"},
                      output_on_backend=True, device=0)
    t_host = time.perf_counter() - t0
    _lib.synchronize(0)
    dt = time.perf_counter() - t0
    print("content-based fuse %s: %.1f ms, %.1f Mvoxels/s (calls queued after %.1f ms)" % (out.shape, dt * 1e3, np.prod(out.shape) / dt / 1e6, t_host * 1e3), flush=True)
print("line launches per fuse():", _lib.get_counter("cb_line_launches", reset=True) / 5, " overflow flag:", _lib.get_counter("cb_overflow", reset=True))
if os.environ.get("MVS_CB_COUNT"):
    print("views", _lib.get_counter("cb_mask_views"), "of them with a box mask", _lib.get_counter("cb_mask_boxes"))
