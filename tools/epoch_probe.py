"""Does the process-to-process spread of register() come from where the lanes' scratch buffers land?  One process, several
epochs: between epochs every pair lane (1..15) is shut down and re-created (its scratch, mailbox and pinned buffers are freed and
allocated again); within an epoch register() runs `reps` times.  python tools/epoch_probe.py [epochs] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multiview_stitcher_amd import _lib, registration
from multiview_stitcher_amd import spatial_image_utils as si

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
grid, tile = np.array([4, 4, 4]), np.array([512, 512, 512])
overlap = np.round(tile * 0.2).astype(int)
tiles, jitters, origins = bench.make_mosaic_on_device(torch, dev, grid, tile, overlap, seed=0, max_jitter=4)[:3]
sims = bench.build_sims(tiles, origins, 0)
torch.cuda.synchronize()
key = si.DEFAULT_TRANSFORM_KEY
import gc
lib = _lib.load()
for e in range(epochs):
    walls = []
    for rep in range(reps + 2):
        t0 = time.perf_counter()
        registration.register(sims, transform_key=key, new_transform_key="reg", device=0, pre_registration_pruning_method="alternating_pattern")
        if rep >= 2:
            walls.append((time.perf_counter() - t0) * 1e3)
        if rep == 1:
            gc.collect(); gc.freeze()
    print(f"epoch {e}: register ms " + " ".join(f"{w:.1f}" for w in walls) + f"  median {np.median(walls):.1f}", flush=True)
    for lane in range(1, 16):
        lib.mvs_shutdown(lane << 8)
