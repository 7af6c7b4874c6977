"""Where the wall time of the 144 pair registrations goes: time inside the library call vs interpreter time per pair."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from multiview_stitcher_amd import _lib, _reg_ops, registration
from multiview_stitcher_amd import spatial_image_utils as si

dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
lib_t, pair_t = [], []
orig_rv, orig_rp = _reg_ops.register_views, registration.register_pair_of_msims

def rv(*a, **k):
    t0 = time.perf_counter()
    try:
        return orig_rv(*a, **k)
    finally:
        lib_t.append(time.perf_counter() - t0)

def rp(*a, **k):
    t0 = time.perf_counter()
    try:
        return orig_rp(*a, **k)
    finally:
        pair_t.append(time.perf_counter() - t0)

_reg_ops.register_views = rv
registration.register_pair_of_msims = rp
for threads in (16, 1):     # (other lane counts: tools/sweep_threads.sh -- a second pool in the same process is not representative)
    for rep in range(3):
        lib_t.clear(); pair_t.clear()
        t0 = time.perf_counter()
        registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pre_registration_pruning_method="alternating_pattern",
                              n_parallel_pairwise_regs=threads)
        wall = (time.perf_counter() - t0) * 1e3
    print(f"threads {threads}: register wall {wall:.1f} ms; pairs {len(pair_t)}; sum pair {sum(pair_t)*1e3:.1f} ms; sum lib {sum(lib_t)*1e3:.1f} ms; "
          f"interpreter per pair {(sum(pair_t)-sum(lib_t))/max(len(pair_t),1)*1e6:.0f} us; lib per pair {sum(lib_t)/max(len(lib_t),1)*1e6:.0f} us")
